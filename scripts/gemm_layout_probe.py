import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from summarizer_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for layout, name, (M, N, K) in ((0, "NT", (12003, 1024, 1024)), (2, "TN", (1024, 1024, 12003)), (2, "TN", (3072, 1024, 12003)), (1, "NN", (12003, 1024, 1024))):
    if layout == 0: a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    elif layout == 1: a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
    else: a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev)
    c = torch.empty(M, N, device=dev)
    for _ in range(3): _lib.check(lib.sumk_gemm_prec(layout, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 0, st), "gemm")
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): lib.sumk_gemm_prec(layout, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 0, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    nb = 2048
    out = np.zeros(nb * 4, dtype=np.uint64)
    lib.sumk_prof_gemm_stamps(out.ctypes.data_as(C.POINTER(C.c_uint64)), nb)
    o = out.reshape(nb, 4).astype(np.float64); o = o[o[:, 3] > 0]
    msg = ""
    if len(o):
        tot, kl, ep, nt = o[:, 0], o[:, 1], o[:, 2], o[:, 3]
        nk = (K + 31) // 32
        msg = f"blocks {len(o)} tiles/block {nt.min():.0f}-{nt.max():.0f} k-loop {np.median(kl/tot)*100:.0f}% cyc/k-tile {np.median(kl/(nt*nk)):.0f} epi/tile {np.median(ep/nt):.0f}"
    print(f"{name} M={M} N={N} K={K}: {us:.1f} us = {2.0*M*N*K/us/1e6:.1f} TF  {msg}")
