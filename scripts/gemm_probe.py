"""Probe: TFLOP/s of the plain fp32 MFMA GEMM entry points on the bench's shapes (interleaved rounds, one process)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from summarizer_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
shapes = [("qkv", 12003, 3072, 1024), ("proj", 12003, 1024, 1024), ("big", 8192, 4096, 4096), ("tn-wgrad", 1024, 1024, 12000)]
bufs = {}
for name, M, N, K in shapes:
    bufs[name] = (torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev))
PREC = 2 if "bf16x6" in sys.argv[1:] else 1 if "bf16x3" in sys.argv[1:] else 0
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(name, M, N, K):
    a, b, c = bufs[name]
    fn = lib.sumk_gemm_tn if name.startswith("tn") else lib.sumk_gemm_nt
    if name.startswith("tn"):
        a = a.t().contiguous() if False else a   # layouts only matter for timing here
    if PREC:
        _lib.check(lib.sumk_gemm_prec(2 if name.startswith("tn") else 0, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, PREC, st), "gemm")
        return
    _lib.check(fn(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, st), "gemm")
res = {n: [] for n, *_ in shapes}
for rnd in range(5):
    for name, M, N, K in shapes:
        for _ in range(2): run(name, M, N, K)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run(name, M, N, K)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        res[name].append(2.0 * M * N * K / dt / 1e12)
for name, M, N, K in shapes:
    r = sorted(res[name])
    print(f"{name:9s} M={M:6d} N={N:5d} K={K:5d}: median {r[len(r)//2]:6.1f} TF/s  (min {r[0]:.1f} max {r[-1]:.1f})")
