#!/usr/bin/env python3
"""Parts of one host -> host streaming step (bench.py --mode stream, 50 videos, 12 003 frames, D = 1024) timed alone on this box:
the native pack into pinned memory per thread count, the H2D copy of the packed batch, the scoring call, the D2H of the scores."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from summarizer_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0"); D = 1024
rng = np.random.default_rng(0)
lens = [int(t) for t in rng.integers(160, 321, size=50)]
arrs = [rng.random((T, D), dtype=np.float32) for T in lens]
n = sum(lens)
host = torch.empty(n, D, dtype=torch.float32).pin_memory(); devx = torch.empty(n, D, dtype=torch.float32, device=dev)
srcs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs]); nrows = np.asarray(lens, dtype=np.int32)
def t(fn, k=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for nt in (0, 4, 8, 16, 32, 64):
    print(f"pack {n * D * 4 / 1e6:.0f} MB, {nt:2d} threads: {t(lambda: lib.sumk_pack_rows(C.c_void_p(host.data_ptr()), srcs, _lib.host_i32(nrows), len(arrs), D, nt)):.3f} ms")
print(f"H2D from pinned: {t(lambda: devx.copy_(host, non_blocking=True)):.3f} ms")
pageable = torch.from_numpy(np.concatenate(arrs))
print(f"H2D from pageable: {t(lambda: devx.copy_(pageable)):.3f} ms")
