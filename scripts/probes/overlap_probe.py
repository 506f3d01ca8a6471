#!/usr/bin/env python3
"""Can an MFMA GEMM run UNDER the persistent BiLSTM recurrence (one cooperative block per CU, latency-bound)?  Times, on the S-TVSum
batch: DSN scoring alone, an input-projection-sized GEMM alone (R x 2048 x 1024), both enqueued on two streams, and both on one."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipes as R
from summarizer_amd import _lib
from summarizer_amd.models.dsn import DSN
lib = _lib.load(); dev = torch.device("cuda:0")
lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)]
torch.manual_seed(1234)
m = DSN(input_size=1024).to(dev).eval()
x = torch.from_numpy(np.concatenate([R.features(T, 1, 1024, i)[:, 0, :] for i, T in enumerate(lens)])).to(dev)
Rr = x.shape[0]
a = torch.randn(Rr, 1024, device=dev); b = torch.randn(2048, 1024, device=dev); c = torch.empty(Rr, 2048, device=dev)
s2 = torch.cuda.Stream()

def gemm(stream):
    _lib.check(lib.sumk_gemm_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), Rr, 2048, 1024, C.c_void_p(stream.cuda_stream)), "gemm")

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    t_dsn = timed(lambda: m.score_packed(x, lens))
    t_gemm = timed(lambda: gemm(torch.cuda.current_stream()))
    def both_two_streams():        # three GEMMs on the side stream: the first runs beside DSN's own projection, the others beside the recurrence
        s2.wait_stream(torch.cuda.current_stream())
        for _ in range(3): gemm(s2)
        m.score_packed(x, lens)
        torch.cuda.current_stream().wait_stream(s2)
    def both_one_stream():
        for _ in range(3): gemm(torch.cuda.current_stream())
        m.score_packed(x, lens)
    t_two = timed(both_two_streams); t_one = timed(both_one_stream)
print(f"DSN scoring alone {t_dsn:.3f} ms; GEMM alone {t_gemm:.3f} ms; same stream {t_one:.3f} ms; two streams {t_two:.3f} ms "
      f"(3 GEMMs beside one scoring call; perfect overlap would be {max(t_dsn, 3 * t_gemm):.3f})")
