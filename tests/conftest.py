import os, sys, json
os.environ.setdefault("SUMK_CHECK", "1")   # verify persistent-kernel health words after every recurrent layer
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def js(arr):
    return json.loads(bytes(arr).decode())


@pytest.fixture(scope="session")
def golden():
    return load_golden
