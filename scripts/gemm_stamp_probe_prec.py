"""Diagnostic: the stamps of gemm_stamp_probe.py for the split-bf16 arithmetics (SUMK_GEMM_DBG=2): cycles per k-tile of the
QKV-shaped NT GEMM against the matrix-pipe floor of 48 / 24 / 8 bf16 MFMAs (32 cycles each) per wave and k-tile."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from summarizer_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 12003, 3072, 1024
a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
for name, prec, mf in (("fp32", 0, 64 * 64), ("bf16x6", 2, 48 * 32), ("bf16x3", 1, 24 * 32), ("bf16", 3, 8 * 32)):
    for _ in range(3):
        _lib.check(lib.sumk_gemm_prec(0, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, prec, st), "gemm")
    torch.cuda.synchronize()
    nb = 2048
    out = np.zeros(nb * 4, dtype=np.uint64)
    _lib.check(lib.sumk_prof_gemm_stamps(out.ctypes.data_as(C.POINTER(C.c_uint64)), nb), "stamps")
    o = out.reshape(nb, 4).astype(np.float64); o = o[o[:, 3] > 0]
    tot, kl, ep, nt = o[:, 0], o[:, 1], o[:, 2], o[:, 3]
    nk = K // 32
    print(f"{name}: {len(o)} blocks ({len(o) / 256:.0f} per CU), tiles/block {nt.min():.0f}-{nt.max():.0f}; block total median {np.median(tot):.0f} cycles; "
          f"k-loop {np.median(kl / tot) * 100:.1f}% epilogue {np.median(ep / tot) * 100:.1f}%; cycles per k-tile {np.median(kl / (nt * nk)):.0f} "
          f"(MFMA pipe floor at this residency: {mf * len(o) / 256:.0f}); epilogue per tile {np.median(ep / nt):.0f}")
