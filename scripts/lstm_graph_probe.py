"""Probe: is the per-step cost host-side (launch rate) or device-side?  Eager vs hipGraph replay of the same call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from summarizer_amd import kernels
from summarizer_amd.models.dsn import DSN
dev = torch.device("cuda:0")
n_seq, H, T, In = 50, 256, 320, 1024
m = DSN(In, H, 1).to(dev).eval()
lens = [T] * n_seq
x = torch.randn(sum(lens), In, device=dev) * 0.1
sb = kernels.SeqBatch.get(lens, dev)
p = dict(m.named_parameters())
with torch.no_grad():
    for _ in range(3):
        kernels.bilstm_layer_forward(x, sb, p, "rnn.", 0, H)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernels.bilstm_layer_forward(x, sb, p, "rnn.", 0, H)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"eager: host returns after {t_host*1e3:.3f} ms, device done after {t_all*1e3:.3f} ms")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        kernels.bilstm_layer_forward(x, sb, p, "rnn.", 0, H)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            h, _ = kernels.bilstm_layer_forward(x, sb, p, "rnn.", 0, H)
        torch.cuda.synchronize()
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"graph replay: {dt*1e3:.3f} ms/call -> {dt/T*1e6:.2f} us/step")
