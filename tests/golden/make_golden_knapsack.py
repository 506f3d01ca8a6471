#!/usr/bin/env python3
"""Knapsack key-shot selection goldens from the REAL reference's `generate_summary(method="knapsack")` + `evaluate_summary`
(summarizer/utils/eval.py:74-165), with the one piece that cannot run here -- OR-tools' solver behind `knapsack_ortools`
(summarizer/utils/knapsack.py:5-23, ortools==7.5.7466, not installable) -- replaced by an EXHAUSTIVE optimal solver that
keeps the reference's own problem statement (values = trunc(1000 * segment mean), weights = frames per segment, capacity =
floor(0.15 * n_frames), knapsack.py:10-15 / eval.py:96-99).

Only instances whose optimal subset is UNIQUE are kept: on those every exact solver -- OR-tools' dynamic-programming solver
included -- must return the same set, so the fixture pins everything around the solver (segment means, the integer
conversion, capacity, the expansion to frames, the F-scores) and the solver itself wherever tie-breaking plays no role.
Run once in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_knapsack.py
"""
import os, sys, types, json
import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
for name in ["h5py", "ortools", "ortools.algorithms", "ortools.algorithms.pywrapknapsack_solver"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["ortools.algorithms"].pywrapknapsack_solver = sys.modules["ortools.algorithms.pywrapknapsack_solver"]
sys.path.insert(0, "/root/reference")

import recipes as R
from summarizer.utils import eval as ref_eval

UNIQUE = {"last": None}


def exhaustive_knapsack(values, weights, items, capacity):
    """Same contract as knapsack_ortools (knapsack.py:5-23); optimal by enumeration of all 2^items subsets."""
    scale = 1000
    v = (np.array(values) * scale).astype(int)           # knapsack.py:13 (np.int there; int == the same truncation)
    w = np.array(weights).astype(int)
    n = int(items)
    assert n <= 24, n
    sv = np.zeros(1, dtype=np.int64); sw = np.zeros(1, dtype=np.int64)
    for i in range(n):                                   # subset sums, subset id = bit mask
        sv = np.concatenate([sv, sv + v[i]]); sw = np.concatenate([sw, sw + w[i]])
    feas = sw <= capacity
    best = sv[feas].max()
    winners = np.flatnonzero(feas & (sv == best))
    UNIQUE["last"] = len(winners) == 1
    m = int(winners[0])
    return [i for i in range(n) if (m >> i) & 1]


ref_eval.knapsack_ortools = exhaustive_knapsack          # eval.py:99 resolves the name in its own module namespace

out, kept, dropped = {}, 0, 0
case = 0
for seed in range(200):
    if kept >= 24:
        break
    rng = np.random.default_rng(7000 + seed)
    T = int(rng.integers(60, 330))
    U = int(rng.integers(3, 19))
    v = R.synthetic_video(T, 7100 + seed, n_users=U)
    # three score styles: uniform noise, peaky, and near-constant (small margins between subsets)
    style = seed % 3
    scores = rng.random(T).astype(np.float32)
    if style == 1:
        scores = (scores ** 4).astype(np.float32)
    elif style == 2:
        scores = (0.5 + 0.05 * scores).astype(np.float32)
    if v["change_points"].shape[0] > 22:
        continue
    summ = ref_eval.generate_summary(scores, v["change_points"], v["n_frames"], v["n_frame_per_seg"].tolist(), v["picks"], 0.15, "knapsack")
    if not UNIQUE["last"]:
        dropped += 1
        continue
    f_avg, f_max = ref_eval.evaluate_summary(summ, v["user_summary"])
    tag = f"c{kept}"
    out[f"{tag}/meta"] = np.array([T, U, 7100 + seed])
    out[f"{tag}/scores"] = scores
    out[f"{tag}/summary"] = summ.astype(np.uint8)
    out[f"{tag}/fscore"] = np.array([f_avg, f_max], dtype=np.float64)
    kept += 1
out["n_cases"] = np.int32(kept)
path = os.path.join(HERE, "knapsack_e2e.npz")
np.savez_compressed(path, **out)
print(f"knapsack_e2e: kept {kept} unique-optimum videos, dropped {dropped} tied ones, {os.path.getsize(path)/1024:.1f} KB")
