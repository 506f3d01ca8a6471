"""Diagnostic: where do a GEMM block's cycles go?  Run with SUMK_GEMM_DBG=2 (and SUMK_GEMM_DMA=0/1): per block the kernel
stamps s_memtime around its k-loops and epilogues; prints medians and the cycles per k-tile against the matrix-pipe floor."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from summarizer_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(12003, 3072, 1024), (12003, 1024, 1024), (8192, 4096, 4096)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
    for _ in range(3): _lib.check(lib.sumk_gemm_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, st), "gemm")
    nb = min(2048, ((M + 127) // 128) * ((N + 127) // 128))
    out = np.zeros(nb * 4, dtype=np.uint64)
    _lib.check(lib.sumk_prof_gemm_stamps(out.ctypes.data_as(C.POINTER(C.c_uint64)), nb), "stamps")
    o = out.reshape(nb, 4).astype(np.float64)
    o = o[o[:, 3] > 0]
    tot, kl, ep, nt = o[:, 0], o[:, 1], o[:, 2], o[:, 3]
    nk = (K + 31) // 32
    print(f"M={M} N={N} K={K}: {len(o)} blocks, tiles/block {nt.min():.0f}-{nt.max():.0f}; block total median {np.median(tot):.0f} max {tot.max():.0f} cycles; "
          f"k-loop {np.median(kl/tot)*100:.1f}% epilogue {np.median(ep/tot)*100:.1f}% other {np.median((tot-kl-ep)/tot)*100:.1f}%; "
          f"cycles per k-tile median {np.median(kl/(nt*nk)):.0f}; epilogue per tile {np.median(ep/nt):.0f}")
