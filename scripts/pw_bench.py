"""Probe: the plane-aware wide GEMM (csrc/gemm_pw.hip) against the in-loop split kernels on the headline shapes.
usage: python scripts/pw_bench.py [reps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from summarizer_amd import kernels, _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
VARIANTS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 10, 1]
STAMPS = "libsumk_diag" in os.environ.get("SUMK_LIB_PATH", "")
lib = _lib.load()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return ts[len(ts) // 2], ts[0]


for (M, N, K) in [(12003, 3072, 1024), (12003, 1024, 1024), (12003, 2048, 1024)]:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev) * 0.03
    c = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    for npl, prec, peak in ((3, 2, 2500.0 / 6), (2, 1, 2500.0 / 3)):
        ap, bp = kernels.split_planes(a, npl), kernels.split_planes(b, npl)
        med, best = timeit(lambda: lib.sumk_gemm_prec(0, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, prec, st))
        print(f"M={M} N={N} K={K} planes={npl}  in-loop split: {med:8.1f} us (best {best:8.1f})  {fl / med / 1e6:7.1f} TF = {fl / med / 1e6 / peak:.3f} of {peak:.0f}")
        for variant in VARIANTS:
            med, best = timeit(lambda: kernels.gemm_planes(ap, M, bp, N, M, N, K, npl, variant=variant, out=c))
            print(f"M={M} N={N} K={K} planes={npl}  gemm_pw var {variant}: {med:8.1f} us (best {best:8.1f})  {fl / med / 1e6:7.1f} TF = {fl / med / 1e6 / peak:.3f} of {peak:.0f}")
        if STAMPS:       # diagnostic build: variant 3 leaves per-block cycle stamps in the front of C
            import numpy as np
            for _ in range(30):
                kernels.gemm_planes(ap, M, bp, N, M, N, K, npl, variant=3, out=c)
            torch.cuda.synchronize()
            st_ = c.view(-1)[:256 * 16].cpu().numpy().view(np.uint64).reshape(256, 8)
            # a grid of fewer than 256 blocks (N = 1024: 252 tiles), or a block whose XCD map names no tile, leaves its slot unwritten -- it then holds
            # product values, not stamps (round 5's record printed "max 1.38e19", "tiles/block 1-1.38e19", "min clock 32" from such slots): keep the
            # slots whose fields are mutually consistent (1 <= tiles <= 64, k-loop + epilogue <= total, a wall time that gives 0.5-3 GHz)
            ok = ((st_[:, 3] >= 1) & (st_[:, 3] <= 64) & (st_[:, 4] > 0) & (st_[:, 1] + st_[:, 2] <= st_[:, 0]) &
                  (st_[:, 0] > 5 * st_[:, 4]) & (st_[:, 0] < 30 * st_[:, 4]))
            st_ = st_[ok]
            print(f"   ({int(ok.sum())} of 256 stamp slots written)")
            if not ok.any():          # (two planes run on gemm_pw16.hip unless variant 32 is asked for: that kernel carries no stamps)
                continue
            tot, loop, epi, nt, rt = (st_[:, i].astype(np.float64) for i in range(5))
            clk = tot / rt * 100.0          # MHz: shader cycles per 100 MHz tick
            steps = nt * (K // 16)
            print(f"   stamps planes={npl}: block total {np.median(tot):.0f} cyc (max {tot.max():.0f}), k-loop {np.median(loop / tot):.3f}, epilogue {np.median(epi / tot):.3f} of it, "
                  f"tiles/block {nt.min():.0f}-{nt.max():.0f}, cycles per k16 step {np.median(loop / steps):.0f} (MFMA floor {36 * 32 * 2 if npl == 3 else 18 * 32 * 2}), "
                  f"epilogue per tile {np.median(epi / nt):.0f} cyc, clock {np.median(clk):.0f} MHz (min {clk.min():.0f}), wall {np.median(rt) / 100:.1f} us")
        med, best = timeit(lambda: kernels.split_planes(a, npl))
        print(f"   split_planes({M} x {K}, {npl}): {med:.1f} us")

# the two MFMA shapes at two planes: gemm_pw16.hip (16x16x32, the product's) against the 32x32x16 plane kernel (variant 32), same operands, interleaved
for (M, N, K) in [(12003, 3072, 1024), (12003, 1024, 1024)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.03
    ap, bp = kernels.split_planes(a, 2), kernels.split_planes(b, 2)
    ref = kernels.gemm_planes(ap, M, bp, N, M, N, K, 2, variant=32)
    got = kernels.gemm_planes(ap, M, bp, N, M, N, K, 2)
    err = float((got - ref).abs().max()); scale = float(ref.abs().max())
    c = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    for variant in (32, 0, 32, 0):
        med, best = timeit(lambda: kernels.gemm_planes(ap, M, bp, N, M, N, K, 2, variant=variant, out=c))
        print(f"M={M} N={N} K={K} planes=2  {'32x32x16' if variant == 32 else '16x16x32'}: {med:8.1f} us (best {best:8.1f})  {fl / med / 1e6:7.1f} TF   [max |d| between the shapes {err:.2e} of {scale:.1f}]")
