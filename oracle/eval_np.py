"""numpy restatement of the evaluation tail (reference: summarizer/utils/eval.py).  TEST INFRASTRUCTURE ONLY.

Deliberately written as slow, literal loops so that it is an independent check of the vectorised
host implementation in summarizer_amd/utils/eval.py.
"""
import math
import numpy as np
from scipy import stats

from .knapsack_np import knapsack_dp


def upsample(scores, n_frames, positions):
    """eval.py:15-35."""
    n_frames = int(n_frames)
    out = np.zeros(n_frames, dtype=np.float32)
    pos = np.asarray(positions)
    if pos.dtype != int:                       # eval.py:25-26
        pos = pos.astype(np.int32)
    if pos[-1] != n_frames:                    # eval.py:27-28
        pos = np.concatenate([pos, [n_frames]])
    for i in range(len(pos) - 1):              # eval.py:29-34
        lo, hi = pos[i], pos[i + 1]
        out[lo:hi] = 0 if i == len(scores) else scores[i]
    return out


def segment_scores(frame_scores, cps):
    """eval.py:88-94: float32 mean over [start, end] inclusive, returned as python floats."""
    seg = []
    for k in range(cps.shape[0]):
        s, e = int(cps[k, 0]), int(cps[k, 1] + 1)
        seg.append(float(frame_scores[s:e].mean()))
    return seg


def generate_summary(scores, cps, n_frames, nfps, positions, proportion=0.15, method="knapsack"):
    """eval.py:74-123."""
    n_segs = cps.shape[0]
    fs = upsample(scores, n_frames, positions)
    seg = segment_scores(fs, cps)
    limits = int(math.floor(n_frames * proportion))       # eval.py:96
    if method == "knapsack":
        picks = knapsack_dp(seg, nfps, n_segs, limits)    # eval.py:99 (OR-tools in the reference; see knapsack_np)
    elif method == "rank":
        order = np.argsort(seg)[::-1].tolist()            # eval.py:101
        picks, total = [], 0
        for i in order:
            if total + nfps[i] < limits:                  # strict '<', eval.py:105
                picks.append(i); total += nfps[i]
    else:
        raise KeyError(f"Unknown method {method}")
    parts = [np.ones(int(nfps[k]), np.float32) if k in picks else np.zeros(int(nfps[k]), np.float32)
             for k in range(n_segs)]                      # eval.py:112-122
    return np.concatenate(parts) if parts else np.zeros(0, np.float32)


def evaluate_summary(machine_summary, user_summary):
    """eval.py:125-165 (float32 arithmetic as under numpy>=2 weak-scalar rules)."""
    m = machine_summary.astype(np.float32).copy()
    u = user_summary.astype(np.float32).copy()
    n_users, n_frames = u.shape
    m[m > 0] = 1
    u[u > 0] = 1
    if len(m) > n_frames:
        m = m[:n_frames]
    elif len(m) < n_frames:
        m = np.concatenate([m, np.zeros(n_frames - len(m))])     # float64 zeros -> m becomes float64 (eval.py:142-143)
    fs = []
    for k in range(n_users):
        gt = u[k]
        ov = (m * gt).sum()
        prec = ov / (m.sum() + 1e-8)
        rec = ov / (gt.sum() + 1e-8)
        fs.append(0.0 if (prec == 0 and rec == 0) else (2 * prec * rec) / (prec + rec))
    return np.mean(fs), np.max(fs)


def evaluate_scores(machine_scores, user_scores, metric="spearmanr"):
    """eval.py:49-72."""
    if metric == "kendalltau":
        fn = lambda x, y: stats.kendalltau(stats.rankdata(-x), stats.rankdata(-y))[0]
    elif metric == "spearmanr":
        fn = lambda x, y: stats.spearmanr(stats.rankdata(-x), stats.rankdata(-y))[0]
    else:
        raise KeyError(f"Unknown metric {metric}")
    return np.mean([fn(machine_scores, user_scores[i]) for i in range(user_scores.shape[0])])
