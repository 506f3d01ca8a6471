"""Probe 2: is the K=1024 shortfall per-tile overhead or tail?  Shapes with an exact number of 128x128 tiles."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from summarizer_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M, N, K in [(4096, 3072, 1024), (4096, 3072, 4096), (8192, 3072, 1024), (12288, 3072, 1024), (12003, 3072, 1024), (4096, 3072, 256), (2048, 3072, 1024)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
    for _ in range(3): lib.sumk_gemm_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, st)
    best = 1e9
    for rnd in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): lib.sumk_gemm_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, st)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"M={M:6d} N={N} K={K:5d} tiles={tiles:5d} ({tiles/768:.2f} x768): {2.0*M*N*K/best/1e12:6.1f} TF/s  {best*1e6:7.1f} us")
