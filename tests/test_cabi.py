"""CPU: the C-ABI library loads and exports every symbol include/sumk.h declares; the product path refuses
to run without the GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re
import subprocess
import pytest
import torch

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "sumk.h")
LIB = os.path.join(ROOT, "summarizer_amd", "libsumk.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"#if 0.*?#endif", "", src, flags=re.S)        # fenced = not yet part of the ABI
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sumk_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in sumk.h but not exported: {missing}"


def test_binding_table_matches_header():
    from summarizer_amd import _lib
    assert sorted(_lib._SIGS) == declared_symbols()


def test_no_cpu_fallback():
    from summarizer_amd._lib import SumkError
    from summarizer_amd.models.vasnet import VASNet
    m = VASNet(input_size=64).eval()
    with torch.no_grad(), pytest.raises(SumkError):
        m(torch.randn(5, 1, 64))


def test_error_reporting_without_gpu(lib):
    lib.sumk_last_error.restype = ctypes.c_char_p
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.sumk_device_count() < 0
    assert b"HIP error" in lib.sumk_last_error()


def test_product_does_not_import_oracle():
    out = subprocess.run(["grep", "-rlE", r"^\s*(from|import)\s+oracle|from \.+oracle", os.path.join(ROOT, "summarizer_amd")],
                         capture_output=True, text=True).stdout.strip()
    assert out == "", f"product files import the oracle: {out}"
