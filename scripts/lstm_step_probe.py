"""Micro-probe: microseconds per recurrence step of sumk_bilstm_layer_forward for several (n_seq, H, T)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from summarizer_amd import kernels
from summarizer_amd.models.dsn import DSN
dev = torch.device("cuda:0")
for n_seq, H, T, In in [(50, 256, 320, 1024), (1, 256, 320, 1024), (32, 256, 320, 1024), (256, 256, 320, 1024), (50, 64, 320, 64), (50, 1024, 100, 1024)]:
    m = DSN(In, H, 1).to(dev).eval()
    lens = [T] * n_seq
    x = torch.randn(sum(lens), In, device=dev) * 0.1
    sb = kernels.SeqBatch.get(lens, dev)
    p = dict(m.named_parameters())
    with torch.no_grad():
        for _ in range(3):
            kernels.bilstm_layer_forward(x, sb, p, "rnn.", 0, H)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            kernels.bilstm_layer_forward(x, sb, p, "rnn.", 0, H)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"n_seq={n_seq:4d} H={H:5d} T={T} In={In}: {dt*1e3:7.3f} ms/call  -> {dt/T*1e6:6.2f} us/step (incl. input GEMM)")
