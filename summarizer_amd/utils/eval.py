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


def evaluate_scores(machine_scores, user_scores, metric="spearmanr"):
    """Compare machine scores with user scores, mean rank correlation over annotators (eval.py:49-72)."""
    n_users, _ = user_scores.shape
    if metric == "kendalltau":
        f = lambda x, y: stats.kendalltau(x, y)[0]
    elif metric == "spearmanr":
        f = lambda x, y: stats.spearmanr(x, y)[0]
    else:
        raise KeyError(f"Unknown metric {metric}")
    rm = stats.rankdata(-machine_scores)                         # ranked once, not once per annotator
    corrs = [f(rm, stats.rankdata(-user_scores[i])) for i in range(n_users)]
    return np.mean(corrs)


def generate_summary(scores, cps, n_frames, nfps, positions, proportion=0.15, method="knapsack"):
    """Generate keyshot-based video summary i.e. a binary vector of shape (sum(nfps),) (eval.py:74-123)."""
    n_segs = cps.shape[0]
    frame_scores = upsample(scores, n_frames, positions)
    seg_score = []
    for seg_idx in range(n_segs):                                # float32 mean per segment, as python floats
        start, end = int(cps[seg_idx, 0]), int(cps[seg_idx, 1] + 1)
        seg_score.append(float(frame_scores[start:end].mean()))
    limits = int(math.floor(n_frames * proportion))
    if method == "knapsack":
        picks = knapsack_ortools(seg_score, nfps, n_segs, limits)
    elif method == "rank":
        order = np.argsort(seg_score)[::-1].tolist()
        picks = []
        total_len = 0
        for i in order:
            if total_len + nfps[i] < limits:                     # strict, eval.py:105
                picks.append(i)
                total_len += nfps[i]
    else:
        raise KeyError(f"Unknown method {method}")
    chosen = np.zeros(n_segs, dtype=np.float32)
    chosen[np.asarray(picks, dtype=np.int64)] = 1
    return np.repeat(chosen, np.asarray(nfps[:n_segs], dtype=np.int64))


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
    f_scores = []
    m_sum = machine_summary.sum()
    for user_idx in range(n_users):
        gt_summary = user_summary[user_idx, :]
        overlap_duration = (machine_summary * gt_summary).sum()
        precision = overlap_duration / (m_sum + 1e-8)
        recall = overlap_duration / (gt_summary.sum() + 1e-8)
        if precision == 0 and recall == 0:
            f_score = 0.
        else:
            f_score = (2 * precision * recall) / (precision + recall)
        f_scores.append(f_score)
    return np.mean(f_scores), np.max(f_scores)
