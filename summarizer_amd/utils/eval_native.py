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


# ---------------------------------------------------------------------------------------------- device-side part (csrc/evaldev.hip)
def device_ready(v):
    """Can this video's tail run on the device?  (ascending picks, <= 4096 of them, <= 32 annotators, change points + ranks present)
    A property of the video's constant metadata: computed once and kept in the prepare_video dict (Trainer.test asks for every video
    of a fold on every call -- 0.5 ms of numpy per 50 videos when recomputed)."""
    r = v.get("_dev_ready")
    if r is None:
        p = v["picks"]
        r = v["_dev_ready"] = bool("cps" in v and "user_ranks" in v and 0 < p.shape[0] <= 4096 and v["user_ranks"].shape[0] <= 32
                                   and bool(np.all(np.diff(p) >= 0)) and v["user_ranks"].shape[1] == v["n_frames"])
    return r


_DEV_BATCH_CACHE = {}      # (ids of the videos' dicts, lens, device) -> the batch's constant descriptors (device + host side); a few entries


def _device_meta(v, device):
    """The video's constant metadata as device tensors, uploaded once and cached inside the prepare_video dict."""
    import torch
    cache = v.setdefault("_dev", {})
    d = cache.get(str(device))
    if d is None:
        ru = v["user_ranks"]
        mu = ru.sum(axis=1) / ru.shape[1]
        d = cache[str(device)] = dict(
            picks=torch.from_numpy(v["picks"]).to(device), cps=torch.from_numpy(v["cps"]).to(device),
            ranks=torch.from_numpy(ru).to(device), mean=torch.from_numpy(mu).to(device),
            ssq=torch.from_numpy(((ru - mu[:, None]) ** 2).sum(axis=1)).to(device))
    return d


def evaluate_batch_device(videos, scores_dev, lens, proportion=0.15, method="knapsack", want_summaries=False, n_threads=0):
    """The same evaluation with the scores still in HBM: `scores_dev` = packed (sum(lens),) float32 device tensor, video i owning rows
    [sum(lens[:i]), sum(lens[:i + 1])).  One launch (a block per video) does upsample + float32 segment means + Spearman on the device
    (sumk_eval_device); one small D2H brings the segment means and correlations home; key-shot selection, summary expansion and
    F-scores finish in the native host threads (sumk_eval_videos with seg_means given).  Same return value as evaluate_batch."""
    import torch
    if method not in METHODS:
        raise KeyError(f"Unknown method {method}")
    lib = _lib.load()
    n, dev = len(videos), scores_dev.device
    if n == 0:
        return np.zeros(0), np.zeros(0), np.zeros(0), ([] if want_summaries else None)
    if not scores_dev.is_cuda or scores_dev.dtype != torch.float32 or not scores_dev.is_contiguous():
        raise _lib.SumkError("evaluate_batch_device: scores must be a contiguous float32 GPU tensor")
    # Everything below that depends only on WHICH videos are scored at WHICH lengths -- the validity checks, the device descriptor block and
    # the constant fields of the host descriptors -- is built once per (videos, lens, device) and kept: a fold's test set is the same on
    # every Trainer.test call.
    key = (tuple(id(v) for v in videos), tuple(int(T) for T in lens), str(dev))
    ent = _DEV_BATCH_CACHE.get(key)
    if ent is None:
        # the kernel's fixed-size LDS tables and its rank rule: the same conditions Trainer._test_on_device tests before coming here
        # (a direct caller past them would corrupt LDS silently: the descriptors live in device memory, sumk_eval_device cannot look)
        for i, (v, T) in enumerate(zip(videos, lens)):
            if not device_ready(v):
                raise _lib.SumkError(f"evaluate_batch_device: video {i} does not qualify for the device tail (needs ascending picks, <= 4096 of "
                                     "them, <= 32 annotators, change points and annotator ranks over n_frames); use evaluate_batch")
            # eval.py:26-34 (upsample): intervals = picks + the n_frames sentinel; one interval per score -- the host tail returns an error
            # for more intervals than scores (csrc/evaltail.hip eval_one), where the reference's loop raises IndexError
            n_picks = v["picks"].shape[0]
            n_int = n_picks - 1 + (1 if v["picks"][-1] != v["n_frames"] else 0)
            if n_int > int(T) + 1:
                raise _lib.SumkError(f"evaluate_batch_device: video {i} has {n_int} pick intervals for {int(T)} scores")
        descr = (_lib.EvalDevVideo * n)()
        metas, row0, frame0, seg0 = [], 0, 0, 0
        for i, (v, T) in enumerate(zip(videos, lens)):
            m = _device_meta(v, dev); metas.append(m)
            e = descr[i]
            e.picks, e.n_picks, e.n_frames, e.n_steps = m["picks"].data_ptr(), v["picks"].shape[0], v["n_frames"], int(T)
            e.row0, e.frame0 = row0, frame0
            e.cps, e.n_segs, e.seg0 = m["cps"].data_ptr(), v["cps"].shape[0], seg0
            e.user_ranks, e.user_mean, e.user_ssq, e.n_users = m["ranks"].data_ptr(), m["mean"].data_ptr(), m["ssq"].data_ptr(), v["user_ranks"].shape[0]
            row0 += int(T); frame0 += v["n_frames"]; seg0 += v["cps"].shape[0]
        descr_dev = torch.frombuffer(bytearray(bytes(descr)), dtype=torch.uint8).to(dev)
        if len(_DEV_BATCH_CACHE) >= 8:
            _DEV_BATCH_CACHE.pop(next(iter(_DEV_BATCH_CACHE)))
        # (the entry holds the video dicts themselves: their ids stay theirs for as long as the entry lives)
        ent = _DEV_BATCH_CACHE[key] = dict(videos=list(videos), metas=metas, descr_dev=descr_dev, rows=row0, frames=frame0, segs=seg0)
    descr_dev, row0, frame0, seg0 = ent["descr_dev"], ent["rows"], ent["frames"], ent["segs"]
    if row0 != scores_dev.numel():
        raise _lib.SumkError(f"evaluate_batch_device: lens sum to {row0}, scores hold {scores_dev.numel()}")
    # Two stages on the device, two small transfers into pinned memory: the segment means come home after the short first kernel and the
    # host's key-shot selection + F-scores run while the device is still correlating (buffers are per batch and reused: the call ends
    # synchronised).
    # (the pinned buffers and events of an entry are shared by every call on that test set: calls from different threads / streams take
    #  turns -- ADVICE r4 -- instead of overwriting each other's results)
    import threading
    with ent.setdefault("lock", threading.Lock()):
        buf = ent.get("buffers")
        if buf is None:
            buf = ent["buffers"] = dict(
                scratch=torch.empty(frame0, dtype=torch.float32, device=dev), seg=torch.empty(max(seg0, 1), dtype=torch.float32, device=dev),
                corr=torch.empty(n, dtype=torch.float64, device=dev),
                part=torch.empty(max(1, lib.sumk_eval_device_spearman_scratch_bytes(n) // 8), dtype=torch.float64, device=dev),
                seg_host=torch.empty(max(seg0, 1), dtype=torch.float32).pin_memory(), corr_host=torch.empty(n, dtype=torch.float64).pin_memory(),
                ev_seg=torch.cuda.Event(), ev_corr=torch.cuda.Event())
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.sumk_eval_device_segments(scores_dev.data_ptr(), descr_dev.data_ptr(), n, buf["scratch"].data_ptr(), buf["seg"].data_ptr(), st),
                   "sumk_eval_device_segments")
        buf["seg_host"].copy_(buf["seg"], non_blocking=True); buf["ev_seg"].record()
        _lib.check(lib.sumk_eval_device_spearman(scores_dev.data_ptr(), descr_dev.data_ptr(), n, buf["part"].data_ptr(), buf["corr"].data_ptr(), st),
                   "sumk_eval_device_spearman")
        buf["corr_host"].copy_(buf["corr"], non_blocking=True); buf["ev_corr"].record()
        seg_means = buf["seg_host"].numpy()[:seg0]       # (a view of the pinned buffer: valid once ev_seg has passed)
        arr = (_lib.EvalVideo * n)()
        summaries, seg_at = [], 0
        try:
            for i, v in enumerate(videos):
                e = arr[i]
                e.n_frames, e.n_steps = v["n_frames"], int(lens[i])
                e.cps, e.nfps, e.n_segs = v["cps"].ctypes.data, v["nfps"].ctypes.data, v["cps"].shape[0]
                e.seg_means = seg_means[seg_at:].ctypes.data
                seg_at += v["cps"].shape[0]
                e.corr = float("nan")             # (passed through untouched when seg_means is given; the device's value is returned below)
                if want_summaries:
                    o = np.empty(int(v["nfps"].sum()), dtype=np.float32); summaries.append(o); e.machine_summary = o.ctypes.data
                if "user_summary" in v:
                    if v["user_summary"].shape[1] != v["n_frames"]:
                        raise ValueError(f"user_summary has {v['user_summary'].shape[1]} frames, video has {v['n_frames']}")
                    e.user_summary, e.n_users = v["user_summary"].ctypes.data, v["user_summary"].shape[0]
            buf["ev_seg"].synchronize()
            _lib.check(lib.sumk_eval_videos(C.cast(arr, C.c_void_p), n, float(proportion), METHODS[method], int(n_threads)), "sumk_eval_videos")
        finally:
            buf["ev_corr"].synchronize()      # the call never returns with its pinned buffers still being written
        corr = buf["corr_host"].numpy().copy()
        f_avg = np.array([arr[i].f_avg for i in range(n)]); f_max = np.array([arr[i].f_max for i in range(n)])
    return corr, f_avg, f_max, (summaries if want_summaries else None)
