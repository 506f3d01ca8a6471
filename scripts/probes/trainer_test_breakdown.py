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

# the host tail alone (key-shot selection + F-scores of the 50 videos, segment means given) for several thread counts
import ctypes as C
from summarizer_amd import _lib
from summarizer_amd.utils import eval_native
lib = _lib.load()
metas = [tr._native_meta(k) for k in keys]
n = len(metas)
seg = [np.random.default_rng(i).random(m["cps"].shape[0]).astype(np.float32) for i, m in enumerate(metas)]
arr = (_lib.EvalVideo * n)()
for i, v in enumerate(metas):
    e = arr[i]
    e.n_frames, e.n_steps = v["n_frames"], 300
    e.cps, e.nfps, e.n_segs = v["cps"].ctypes.data, v["nfps"].ctypes.data, v["cps"].shape[0]
    e.seg_means = seg[i].ctypes.data
    e.user_summary, e.n_users = v["user_summary"].ctypes.data, v["user_summary"].shape[0]
for nt in (0, 1, 2, 4, 8, 16, 32):        # 0 = the library's persistent worker pool (the default)
    for _ in range(3): lib.sumk_eval_videos(C.cast(arr, C.c_void_p), n, 0.15, 0, nt)
    t0 = time.perf_counter()
    for _ in range(50): lib.sumk_eval_videos(C.cast(arr, C.c_void_p), n, 0.15, 0, nt)
    print(f"host tail, {nt:2d} threads: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
print("hardware threads", os.cpu_count())
