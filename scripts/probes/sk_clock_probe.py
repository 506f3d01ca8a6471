#!/usr/bin/env python3
"""At what shader clock does ONE video's QKV-sized small-batch launch (300 x 3072 x 1024, the SK instance of csrc/gemm_lean.hip) run when it is
issued the way one-video-per-call scoring issues it (a ~100 us cadence of small kernels), and how many cycles are its k-loops?  Diagnostic
build: SUMK_LIB_PATH=summarizer_amd/libsumk_diag.so SUMK_GEMM_DBG=2.  Per block the kernel stamps s_memtime (shader cycles) and
s_memrealtime (100 MHz) at its start and end."""
import os, sys, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from summarizer_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 300, 3072, 1024
a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
nb = lib.sumk_gemm_splitk_workspace_bytes(M, N, 1)
ws = torch.zeros(nb + 256, dtype=torch.uint8, device=dev); wsp = (ws.data_ptr() + 255) // 256 * 256
filler = torch.randn(300, 1024, device=dev)
def call():
    _lib.check(lib.sumk_gemm_splitk(0, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, K, K, N, 1, 0, None, N, None, 1.0, wsp, nb, st), "splitk")
for mode in ("back to back", "with ~70 us of small kernels between calls"):
    for _ in range(300):
        call()
        if mode != "back to back":
            for _ in range(8): filler.mul_(1.0)
    torch.cuda.synchronize()
    nblk = 240
    out = np.zeros(nblk * 4, dtype=np.uint64); fine = np.zeros(nblk * 8, dtype=np.uint64)
    _lib.check(lib.sumk_prof_gemm_stamps(out.ctypes.data_as(C.POINTER(C.c_uint64)), nblk), "stamps")
    _lib.check(lib.sumk_prof_gemm_stamps(fine.ctypes.data_as(C.POINTER(C.c_uint64)), -nblk), "stamps")
    o = out.reshape(nblk, 4).astype(np.float64); f = fine.reshape(nblk, 8).astype(np.float64)
    ok = o[:, 3] > 0
    tot, kl = o[ok, 0], o[ok, 1]
    real_us = (f[ok, 5] - f[ok, 4]) / 100.0
    print(f"{mode}: block total {np.median(tot):.0f} cycles (k-loops {np.median(kl):.0f}) in {np.median(real_us):.2f} us -> shader clock {np.median(tot / real_us) / 1e3:.2f} GHz; "
          f"k-loop floor 512 MFMAs x 64 = 32768 cycles")
