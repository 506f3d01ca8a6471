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


@pytest.fixture(autouse=True)
def _poisoned_scratch(monkeypatch):
    """SUMK_TEST_POISON=1 python -m pytest tests -m gpu: EVERY test runs with uninitialised device allocations handed out as 0xFF bytes (NaN as
    fp32 / bf16, -1 as an integer) -- a kernel that reads scratch nobody wrote then fails its test's own gates.  Off by default (the
    dedicated tests/test_gpu_poison.py always runs); the whole-suite pass is an occasional audit."""
    if os.environ.get("SUMK_TEST_POISON") != "1":
        yield
        return
    import torch
    from summarizer_amd import kernels
    empty, empty_like, ws = torch.empty, torch.empty_like, kernels.workspace

    def fill(t):
        if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() and t.is_contiguous():
            t.reshape(-1).view(torch.uint8).fill_(255)
        return t
    monkeypatch.setattr(torch, "empty", lambda *a, **k: fill(empty(*a, **k)))
    monkeypatch.setattr(torch, "empty_like", lambda *a, **k: fill(empty_like(*a, **k)))
    monkeypatch.setattr(kernels, "workspace", lambda *a, **k: fill(ws(*a, **k)))
    yield
