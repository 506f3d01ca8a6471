"""Probe: does the row stride of the operands (lda = K for the plain entry points) matter?  Power-of-two strides put the same
k-chunk of every row on the same L2 channel; compares K = 1024 with neighbours that are not powers of two."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from summarizer_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N = 12003, 3072
for K in (1024, 992, 1056, 1040, 1088, 2048, 2080, 4096, 4128):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
    for _ in range(3): _lib.check(lib.sumk_gemm_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, st), "gemm")
    best = 0
    for rnd in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): lib.sumk_gemm_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        best = max(best, 2.0 * M * N * K / dt / 1e12)
    print(f"M={M} N={N} K={K:5d} (row stride {4*K:6d} B): best {best:6.1f} TF/s", flush=True)
