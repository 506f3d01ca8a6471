"""Evaluation tail -- mirror of `summarizer/utils/eval.py` (same function names, arguments, return values and
error behaviour), vectorised where that cannot change a bit of the result.  CPU code in the reference too
(SURVEY.md section 8a rows a10-a14); the knapsack runs in native code (utils/knapsack.py -> libsumk.so)."""
import math
import numpy as np
from scipy import stats

from .knapsack import knapsack_ortools


def upsample(scores, n_frames, positions):
    """Upsample scores vector to the original number of frames (eval.py:15-35): piecewise constant between
    consecutive `positions`, zero before positions[0] and for an interval past the last score."""
    n_frames = int(n_frames)
    positions = np.asarray(positions)
    if positions.dtype != int:                                   # eval.py:25-26
        positions = positions.astype(np.int32)
    if positions[-1] != n_frames:                                # eval.py:27-28
        positions = np.concatenate([positions, [n_frames]])
    frame_scores = np.zeros((n_frames), dtype=np.float32)
    n_int = len(positions) - 1
    vals = np.zeros(n_int, dtype=np.float32)
    k = min(n_int, len(scores))
    vals[:k] = np.asarray(scores, dtype=np.float32)[:k]          # interval i == len(scores) gets 0 (eval.py:31-32)
    if n_int > len(scores) + 1:
        raise IndexError(f"index {len(scores) + 1} is out of bounds for axis 0 with size {len(scores)}")   # scores[i] in the reference
    lo = np.clip(positions[:-1], 0, n_frames)
    hi = np.clip(positions[1:], 0, n_frames)
    if np.all(hi[:-1] <= lo[1:]) and np.all(lo <= hi):
        # monotone positions: one repeat; identical to the reference's slice assignments
        lens = hi - lo
        frame_scores[lo[0]:lo[0] + lens.sum()] = np.repeat(vals, lens) if np.all(hi[:-1] == lo[1:]) else 0
        if not np.all(hi[:-1] == lo[1:]):
            for i in range(n_int):
                frame_scores[lo[i]:hi[i]] = vals[i]
    else:
        for i in range(n_int):                                    # arbitrary (overlapping) positions: literal order
            frame_scores[positions[i]:positions[i + 1]] = vals[i]
    return frame_scores


def generate_scores(probs, n_frames, positions):
    """Set score to every original frame of the video for comparison with annotations (eval.py:37-47)."""
    return upsample(probs, n_frames, positions)


def evaluate_scores(machine_scores, user_scores, metric="spearmanr", user_ranks=None):
    """Compare machine scores with user scores, mean rank correlation over annotators (eval.py:49-72).

    The reference calls `stats.spearmanr(rankdata(-x), rankdata(-y))` per annotator, i.e. it ranks the (constant)
    annotator scores again for every video of every evaluation and lets spearmanr re-rank both inputs.  Here the machine
    ranks are computed once per video, annotator ranks can be passed in precomputed (`user_ranks`, see `rank_users`), and
    Spearman's rho is the Pearson correlation of the two rank vectors (float64, like scipy) -- the same value up to
    floating-point summation order (~1e-15)."""
    n_users, _ = user_scores.shape
    if metric == "kendalltau":
        rm = stats.rankdata(-machine_scores)
        corrs = [stats.kendalltau(rm, stats.rankdata(-user_scores[i]))[0] for i in range(n_users)]
        return np.mean(corrs)
    if metric != "spearmanr":
        raise KeyError(f"Unknown metric {metric}")
    rm = stats.rankdata(-machine_scores)
    ru = rank_users(user_scores) if user_ranks is None else user_ranks
    rm = rm - rm.mean()
    ru = ru - ru.mean(axis=1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        corrs = (ru @ rm) / np.sqrt((ru * ru).sum(axis=1) * (rm @ rm))   # NaN for a constant vector, like scipy
    return np.mean(corrs)


def rank_users(user_scores):
    """rankdata(-user_scores[i]) for every annotator, stacked (n_users, n_frames) float64: constant per video, so
    callers (Trainer._eval_scores) compute it once and reuse it at every evaluation."""
    return np.stack([stats.rankdata(-user_scores[i]) for i in range(user_scores.shape[0])])


def _segment_scores(frame_scores, cps):
    """float32 mean of the frame scores over each [start, end] change-point interval, as python floats (eval.py:88-94)."""
    return [float(frame_scores[int(lo):int(hi) + 1].mean()) for lo, hi in cps[:, :2]]


def _select_segments(seg_score, nfps, n_segs, budget, method):
    """Indices of the segments kept under the frame budget: 0/1 knapsack (native DP) or greedy by score (eval.py:98-110)."""
    if method == "knapsack":
        return knapsack_ortools(seg_score, nfps, n_segs, budget)
    if method != "rank":
        raise KeyError(f"Unknown method {method}")
    kept, used = [], 0
    for i in np.argsort(seg_score)[::-1].tolist():
        if used + nfps[i] < budget:                 # strict, eval.py:105
            kept.append(i)
            used += nfps[i]
    return kept


def generate_summary(scores, cps, n_frames, nfps, positions, proportion=0.15, method="knapsack"):
    """Keyshot-based video summary: binary float32 vector of length sum(nfps) (eval.py:74-123).
    scores: per-step importance; cps: (n_segs, 2) change points; nfps: frames per segment; positions: sampled-frame
    positions; proportion: summary length budget; method: 'knapsack' | 'rank'."""
    n_segs = cps.shape[0]
    seg_score = _segment_scores(upsample(scores, n_frames, positions), cps)
    kept = _select_segments(seg_score, nfps, n_segs, int(math.floor(n_frames * proportion)), method)
    flags = np.zeros(n_segs, dtype=np.float32)
    flags[np.asarray(kept, dtype=np.int64)] = 1
    return np.repeat(flags, np.asarray(nfps[:n_segs], dtype=np.int64))


def evaluate_summary(machine_summary, user_summary):
    """Compare machine summary with user summary (keyshot-based): (avg, max) F-score over annotators (eval.py:125-165)."""
    machine_summary = machine_summary.astype(np.float32)
    user_summary = user_summary.astype(np.float32)
    n_users, n_frames = user_summary.shape
    machine_summary[machine_summary > 0] = 1
    user_summary[user_summary > 0] = 1
    if len(machine_summary) > n_frames:
        machine_summary = machine_summary[:n_frames]
    elif len(machine_summary) < n_frames:
        zero_padding = np.zeros((n_frames - len(machine_summary)))            # float64 on purpose (eval.py:142)
        machine_summary = np.concatenate([machine_summary, zero_padding])
    # all annotators at once; every operation below is the reference's, elementwise, in the same dtype
    # (float32, or float64 after the float64 zero padding above), and the sums are sums of 0/1 values -- exact in any
    # order -- so the result is bit-identical to the per-annotator loop of eval.py:149-160.
    m_sum = machine_summary.sum()
    overlap = (user_summary * machine_summary[None, :]).sum(axis=1)
    precision = overlap / (m_sum + 1e-8)
    recall = overlap / (user_summary.sum(axis=1) + 1e-8)
    with np.errstate(invalid="ignore", divide="ignore"):
        f_scores = (2 * precision * recall) / (precision + recall)
    f_scores = np.where((precision == 0) & (recall == 0), 0., f_scores)
    return np.mean(f_scores), np.max(f_scores)
