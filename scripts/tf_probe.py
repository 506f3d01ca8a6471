# Probe: per-call wall time of the Transformer scorer by precision (fp32 / bf16x6 / bf16x3 interleaved), then the bench legs.
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import bench
from summarizer_amd.models.transformer import Transformer
dev = torch.device("cuda:0")
lens = bench.tvsum_lens(50); frames = int(sum(lens))
x = torch.randn(frames, 1024, device=dev) * 0.05
torch.manual_seed(1234)
m = Transformer(input_size=1024).to(dev).eval()
for prec in ("fp32", "bf16x6", "fp32", "bf16x3", "fp32"):
    m.precision = prec
    ts = []
    with torch.no_grad():
        for _ in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s = m.score_packed(x, lens)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(prec, " ".join(f"{t:.2f}" for t in ts))
print(bench.transformer_legs(x, lens, dev, frames))
