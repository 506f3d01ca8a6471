"""Probe: latency of ONE video per call (the reference's own calling pattern, B = 1, T = 300, D = 1024)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from summarizer_amd.models.vasnet import VASNet
from summarizer_amd.models.dsn import DSN
from summarizer_amd.training import FlatAdam
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(300, 1, 1024, device=dev) * 0.5
tgt = torch.rand(300, 1, 1, device=dev)
for name, m in (("vasnet", VASNet()), ("dsn", DSN())):
    m = m.to(dev).eval()
    with torch.no_grad():
        for _ in range(20): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"{name}: score one video  {dt*1e6:8.1f} us  -> {300/dt/1e6:.2f} M frames/s")
    m.train()
    opt = FlatAdam(m.parameters(), lr=5e-5, weight_decay=1e-5)
    def step():
        opt.zero_grad()
        loss = torch.mean((m(x) - tgt) ** 2)
        loss.backward()
        opt.step()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print(f"{name}: train one video  {dt*1e6:8.1f} us  -> {300/dt/1e6:.3f} M frames/s")
