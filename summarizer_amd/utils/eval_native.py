"""Batch evaluation tail in native threads (libsumk.so: `sumk_eval_videos`, csrc/evaltail.hip): for a list of videos it
does what `utils/eval.py` does one video at a time in numpy -- upsample, key-shot summary (knapsack / rank), F-scores
against the annotators and the Spearman correlation -- with bit-identical summaries and F-scores (tests/test_host_eval.py)
and the correlation equal to ~1e-15.  This is what `Trainer.test` / `predict_dataset` call; the numpy functions stay the
readable specification (and the fallback for `metric="kendalltau"`)."""
import ctypes as C
import numpy as np

from .. import _lib

METHODS = {"knapsack": 0, "rank": 1}


def prepare_video(n_frames, picks, cps=None, nfps=None, user_summary=None, user_ranks=None):
    """Contiguous, C-typed copies of one video's constant evaluation metadata (done once per video, then cached)."""
    picks = np.asarray(picks)
    if picks.dtype != int:                                   # eval.py:25-26
        picks = picks.astype(np.int32)
    d = dict(n_frames=int(n_frames), picks=np.ascontiguousarray(picks, dtype=np.int32))
    if cps is not None:
        d["cps"] = np.ascontiguousarray(np.asarray(cps)[:, :2], dtype=np.int32)
        d["nfps"] = np.ascontiguousarray(np.asarray(nfps), dtype=np.int32)
    if user_summary is not None:
        d["user_summary"] = np.ascontiguousarray(user_summary, dtype=np.float32)
    if user_ranks is not None:
        d["user_ranks"] = np.ascontiguousarray(user_ranks, dtype=np.float64)
    return d


def evaluate_batch(videos, scores, proportion=0.15, method="knapsack", want_summaries=False, n_threads=0):
    """videos: list of `prepare_video` dicts; scores: list of (n_steps,) float32 arrays.
    Returns (corr (n,), f_avg (n,), f_max (n,), summaries or None); entries are NaN where the inputs were not given."""
    if method not in METHODS:
        raise KeyError(f"Unknown method {method}")
    lib = _lib.load()
    n = len(videos)
    arr = (_lib.EvalVideo * max(n, 1))()
    keep, summaries = [], []
    for i, (v, s) in enumerate(zip(videos, scores)):
        s = np.ascontiguousarray(np.atleast_1d(s), dtype=np.float32)
        keep.append(s)
        e = arr[i]
        e.scores, e.n_steps = s.ctypes.data, s.shape[0]
        e.picks, e.n_picks, e.n_frames = v["picks"].ctypes.data, v["picks"].shape[0], v["n_frames"]
        if "cps" in v:
            e.cps, e.nfps, e.n_segs = v["cps"].ctypes.data, v["nfps"].ctypes.data, v["cps"].shape[0]
            if want_summaries:
                out = np.empty(int(v["nfps"].sum()), dtype=np.float32)
                summaries.append(out)
                e.machine_summary = out.ctypes.data
        if "user_summary" in v and "cps" in v:
            if v["user_summary"].shape[1] != v["n_frames"]:
                raise ValueError(f"user_summary has {v['user_summary'].shape[1]} frames, video has {v['n_frames']}")
            e.user_summary, e.n_users = v["user_summary"].ctypes.data, v["user_summary"].shape[0]
        if "user_ranks" in v:
            e.user_ranks = v["user_ranks"].ctypes.data
            e.n_users = v["user_ranks"].shape[0] if e.n_users == 0 else e.n_users
    _lib.check(lib.sumk_eval_videos(C.cast(arr, C.c_void_p), n, float(proportion), METHODS[method], int(n_threads)),
               "sumk_eval_videos")
    corr = np.array([arr[i].corr for i in range(n)]); f_avg = np.array([arr[i].f_avg for i in range(n)])
    f_max = np.array([arr[i].f_max for i in range(n)])
    return corr, f_avg, f_max, (summaries if want_summaries else None)
