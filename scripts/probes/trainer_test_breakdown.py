#!/usr/bin/env python3
"""Where a Trainer.test call on 50 videos spends its wall time (bench.py trainer_test_mode): cProfile of 30 calls + a timed split of
_test_on_device into pack / score / device tail + D2H / host tail."""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from summarizer_amd.models.vasnet import VASNetTrainer
from summarizer_amd.utils.datasets import synthetic_dataset
from summarizer_amd.utils.hps import make_hps
ds = synthetic_dataset(50, seed=11, D=1024, t_range=(150, 320), n_users=20)
keys = list(ds.keys())
hps = make_hps(ds, [{"train_keys": [], "test_keys": keys}], epochs=1, extra_params={})
torch.manual_seed(1234)
tr = VASNetTrainer(hps, hps.splits_files[0]).reset()
for _ in range(5): tr.test(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): tr.test(0)
torch.cuda.synchronize()
print(f"Trainer.test: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(30): tr.test(0)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
