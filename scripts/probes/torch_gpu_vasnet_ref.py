#!/usr/bin/env python3
"""Yardstick, not product: the VASNet scoring arithmetic (vasnet.py:114-145: three projections, Q.K^T * scale, softmax, alpha.V, output
projection + residual, LayerNorm, Linear + ReLU, LayerNorm, Linear + sigmoid) as plain PyTorch-ROCm fp32 ops on this GPU, (a) one video per
call in a Python loop -- how the reference's own code would run after `.cuda()` -- and (b) the vendor GEMM (rocBLAS / hipBLASLt fp32) on the
three large projections of the packed batch.  Same synthetic S-TVSum shapes as bench.py (50 videos, 12 003 frames, D = 1024)."""
import time, numpy as np, torch
import torch.nn.functional as F
dev = torch.device("cuda:0"); D = 1024
rng = np.random.default_rng(0)
lens = [int(t) for t in rng.integers(160, 321, size=50)]; lens[-1] += 12003 - sum(lens)
g = torch.Generator(device="cpu"); g.manual_seed(0)
W = {k: (torch.randn(D, D, generator=g) / 32).to(dev) for k in ("q", "k", "v", "o", "1")}
b1 = torch.zeros(D, device=dev); w2 = (torch.randn(1, D, generator=g) / 32).to(dev); b2 = torch.zeros(1, device=dev)
lw = torch.ones(D, device=dev); lb = torch.zeros(D, device=dev)
xs = [torch.randn(T, D, generator=g).to(dev) for T in lens]

def score(x):
    K, Q, V = x @ W["k"].t(), x @ W["q"].t(), x @ W["v"].t()
    a = torch.softmax((Q @ K.t()) * 0.06, dim=-1)
    y = F.layer_norm((a @ V) @ W["o"].t() + x, (D,), lw, lb, 1e-6)
    y = F.layer_norm(torch.relu(y @ W["1"].t() + b1), (D,), lw, lb, 1e-6)
    return torch.sigmoid(y @ w2.t() + b2)

def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

with torch.no_grad():
    dt = timed(lambda: [score(x) for x in xs], 20)
    print(f"PyTorch-ROCm fp32, one video per call, 50 videos: {dt * 1e3:.3f} ms = {dt / 50 * 1e6:.1f} us per video, {sum(lens) / dt / 1e6:.2f} M frames/s")
    x300 = torch.randn(300, D, generator=g).to(dev)
    dt1 = timed(lambda: score(x300), 200)
    print(f"PyTorch-ROCm fp32, one T = 300 video: {dt1 * 1e6:.1f} us")
    xp = torch.cat(xs); wqkv = torch.cat([W["q"], W["k"], W["v"]])
    for name, fn, flop in (("QKV  (R,D)x(3D,D)^T fp32", lambda: xp @ wqkv.t(), 2 * 12003 * D * 3 * D), ("proj (R,D)x(D,D)^T fp32", lambda: xp @ W["o"].t(), 2 * 12003 * D * D)):
        dtg = timed(fn, 50)
        print(f"{name}: {dtg * 1e6:.1f} us, {flop / dtg / 1e12:.1f} TFLOP/s = {flop / dtg / 157.3e12:.2f} of the fp32 MFMA peak")
