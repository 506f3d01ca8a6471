"""Fixed vs per-k-tile cost of the bf16-source GEMM kernels: (R x N) = A(R x K) B(N x K)^T for K = 64 ... 4096 (SUMK_B16_WIDE picks the tile)."""
import ctypes as C
import sys
import torch
from summarizer_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
R = 12003
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for K in (64, 256, 1024, 2048, 4096):
    a = torch.randn(R, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.zeros(R, N, device=dev)
    fn = lambda: _lib.check(lib.sumk_gemm_bf16src(0, a.data_ptr(), b.data_ptr(), c.data_ptr(), R, N, K, None, 0, st), "b16")
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e3
    print(f"N={N} K={K:5d}: {t:7.1f} us  {2.0 * R * N * K / t / 1e6:7.1f} TF/s", flush=True)
