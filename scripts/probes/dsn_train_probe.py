"""Probe of the DSN training legs on the S-TVSum batch: MSE step and REINFORCE step (bench.py's own step functions), ms per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from summarizer_amd import kernels
from summarizer_amd.models.dsn import DSN

dev = torch.device("cuda:0")
lens = bench.tvsum_lens(50)
frames = sum(lens)
torch.manual_seed(1234)
x = torch.randn(frames, 1024, device=dev) * 0.5
m = DSN(input_size=1024).to(dev).train()
step, _opt = bench.make_reinforce_step(m, x, lens, dev)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"REINFORCE step R6={os.environ.get('SUMK_LSTM_BWD_R6', '1')}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms", flush=True)
kernels.health_check()
