"""Probe: BASELINE config 5 (8 x (10 000, 2048)) scoring in bf16x6, a few steps (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from summarizer_amd.models.vasnet import VASNet
dev = torch.device("cuda:0")
D, lens = 2048, [10000] * int(os.environ.get("NSEQ", "8"))
torch.manual_seed(1234)
m = VASNet(input_size=D, precision=os.environ.get("PREC", "bf16x6")).to(dev).eval()
g = torch.Generator(device=dev); g.manual_seed(0)
x = torch.randn(sum(lens), D, device=dev, generator=g) * 0.05
with torch.no_grad():
    s = m.score_packed(x, lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s = m.score_packed(x, lens)
    torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per step; finite {bool(torch.isfinite(s).all())}")
