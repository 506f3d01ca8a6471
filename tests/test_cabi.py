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


def test_plane_size_queries_need_no_gpu(lib):
    """The size queries of the round-5 plane entry points are pure host arithmetic: they answer (and refuse ineligible shapes with 0) without a GPU."""
    import ctypes as C
    lib.sumk_planes_bytes.restype = C.c_size_t; lib.sumk_planes_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.sumk_vasnet_wplanes_bytes.restype = C.c_size_t; lib.sumk_vasnet_wplanes_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.sumk_bilstm_wplanes_bytes.restype = C.c_size_t; lib.sumk_bilstm_wplanes_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.sumk_attn_planes_alpha_bytes.restype = C.c_size_t; lib.sumk_attn_planes_alpha_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    # rows are padded to 64; 2 bytes per element and plane; 8 KiB of slack a row tile may read past the last sub-array
    assert lib.sumk_planes_bytes(12003, 1024, 3) == 12032 * 1024 * 3 * 2 + 8192
    assert lib.sumk_planes_bytes(64, 16, 2) == 64 * 16 * 2 * 2 + 8192
    for bad in ((0, 1024, 3), (100, 24, 3), (100, 1024, 1), (100, 1024, 4), (100, 8, 2)):
        assert lib.sumk_planes_bytes(*bad) == 0, bad
    assert lib.sumk_vasnet_wplanes_bytes(1024, 3) > 5 * 1024 * 1024 * 6 and lib.sumk_vasnet_wplanes_bytes(1024, 2) > 5 * 1024 * 1024 * 4
    for bad in ((1000, 3), (128, 3), (1024, 1), (1024, 4), (0, 3)):
        assert lib.sumk_vasnet_wplanes_bytes(*bad) == 0, bad
    assert lib.sumk_bilstm_wplanes_bytes(1024, 256, 3) >= 2048 * 1024 * 6 + 2048 * 4
    for bad in ((1024, 100, 3), (1000, 256, 3), (64, 256, 3), (1024, 256, 5)):
        assert lib.sumk_bilstm_wplanes_bytes(*bad) == 0, bad
    lib.sumk_transformer_wplanes_bytes.restype = C.c_size_t; lib.sumk_transformer_wplanes_bytes.argtypes = [C.c_int32] * 4
    per_layer = 6 * 1024 * 1024 * 6          # in_proj (3 D^2) + out_proj + linear1 + linear2 at three planes, 2 bytes each
    assert lib.sumk_transformer_wplanes_bytes(1024, 1024, 6, 3) >= 6 * per_layer + 1024 * 1024 * 6
    assert lib.sumk_transformer_wplanes_bytes(1024, 1024, 6, 3) < 6 * per_layer + 1024 * 1024 * 6 + 64 * 9000
    for bad in ((1000, 1024, 6, 3), (1024, 1000, 6, 3), (1024, 1024, 0, 3), (1024, 1024, 6, 1), (128, 128, 2, 2)):
        assert lib.sumk_transformer_wplanes_bytes(*bad) == 0, bad
    # the inference workspace grows by the plane buffers only in the split-bf16 precisions, only in inference, only for eligible shapes
    lib.sumk_transformer_workspace_bytes.restype = C.c_size_t; lib.sumk_transformer_workspace_bytes_for.restype = C.c_size_t
    off = (C.c_int32 * 3)(0, 200, 500)
    base = lib.sumk_transformer_workspace_bytes(1024, 1024, 8, 2, 2, off, 0)
    assert base > 0 and lib.sumk_transformer_workspace_bytes_for(1024, 1024, 8, 2, 2, off, 0, 0) == base
    assert lib.sumk_transformer_workspace_bytes_for(1024, 1024, 8, 2, 2, off, 0, 2) > base + 2 * 512 * 1024 * 6        # bf16x6: two activation-plane buffers at least
    assert lib.sumk_transformer_workspace_bytes_for(1024, 1024, 8, 2, 2, off, 1, 2) == lib.sumk_transformer_workspace_bytes(1024, 1024, 8, 2, 2, off, 1)
    assert lib.sumk_transformer_workspace_bytes_for(64, 64, 4, 2, 2, off, 0, 2) == lib.sumk_transformer_workspace_bytes(64, 64, 4, 2, 2, off, 0)
    assert lib.sumk_attn_planes_alpha_bytes(12003, 320, 3) == 12032 * 320 * 3 * 2 + 8192
    assert lib.sumk_attn_planes_alpha_bytes(12003, 300, 2) == 12032 * 320 * 2 * 2 + 8192        # key count rounded up to 32
    assert lib.sumk_attn_planes_alpha_bytes(0, 320, 3) == 0 and lib.sumk_attn_planes_alpha_bytes(10, 320, 1) == 0


def test_hot_kernels_compile_without_waterfall_loops(tmp_path):
    """The compiler wraps a buffer instruction in a WATERFALL loop (v_readfirstlane / v_cmp_eq / s_and_saveexec ... s_cbranch_execnz) when it cannot prove
    the instruction's descriptor or scalar offset wave-uniform.  Round 6 found such loops around every LDS-DMA instruction of a re-cut attention launch
    (block coordinates out of shuffles), around the MC-operand loads of the wide bf16 GEMM (k-tile strides carried through the tile loop) and around the 32
    exchange loads per step of the sLSTM recurrence (wave id left in a VGPR) -- DESIGN.md section 3, "reading the compiler's output".  The four sources
    they lived in must compile to gfx950 assembly without one (CPU only: hipcc cross-compiles; scripts/check_waterfalls.sh does every source)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "summarizer_amd", "csrc")
    names = ["attn_pw", "attn_b16", "gemm_b16", "lstm"]
    procs = [(n, subprocess.Popen([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", n + ".hip", "-o", str(tmp_path / (n + ".s"))],
                                  cwd=csrc, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)) for n in names]
    for n, p in procs:
        assert p.wait(timeout=900) == 0, n
        asm = open(tmp_path / (n + ".s")).read()
        assert "v_mfma" in asm, n
        assert asm.count("s_and_saveexec_b64 vcc, vcc") == 0, (n, asm.count("s_and_saveexec_b64 vcc, vcc"))
