"""CPU: host-side evaluation tail (summarizer_amd/utils/eval.py + native knapsack) vs golden vectors from the real
reference and vs the literal oracle.  Bit-exact for the integer / index work (upsample, summaries, picks)."""
import numpy as np
import pytest

import recipes as R
from conftest import load_golden
from oracle import eval_np, knapsack_np
from summarizer_amd.utils import eval as E
from summarizer_amd.utils.knapsack import knapsack_ortools


def test_metrics_vs_reference_goldens():
    g = load_golden("metrics")
    for ci in range(3):
        T, U, seed = g[f"c{ci}/T_U_seed"]
        v = R.synthetic_video(int(T), int(seed), n_users=int(U))
        sc = g[f"c{ci}/scores"]
        fs = E.upsample(sc, v["n_frames"], v["picks"])
        np.testing.assert_array_equal(fs, g[f"c{ci}/frame_scores"])
        assert fs.dtype == np.float32
        summ = E.generate_summary(sc, v["change_points"], v["n_frames"], v["n_frame_per_seg"].tolist(), v["picks"], 0.15, "rank")
        np.testing.assert_array_equal(summ, g[f"c{ci}/summary_rank"])
        assert summ.dtype == np.float32
        np.testing.assert_array_equal(np.array(E.evaluate_summary(summ, v["user_summary"])), g[f"c{ci}/fscore"])
        np.testing.assert_array_equal(np.array(E.evaluate_summary(summ[:-7], v["user_summary"])), g[f"c{ci}/fscore_short"])
        long = np.concatenate([summ, np.ones(5, np.float32)])
        np.testing.assert_array_equal(np.array(E.evaluate_summary(long, v["user_summary"])), g[f"c{ci}/fscore_long"])
        np.testing.assert_allclose(E.evaluate_scores(fs, v["user_scores"]), g[f"c{ci}/spearman"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(E.evaluate_scores(fs, v["user_scores"], user_ranks=E.rank_users(v["user_scores"])),
                                   g[f"c{ci}/spearman"], rtol=1e-12, atol=1e-15)
        if np.isfinite(g[f"c{ci}/kendall"]):
            assert E.evaluate_scores(fs, v["user_scores"], metric="kendalltau") == g[f"c{ci}/kendall"]
    with pytest.raises(KeyError):
        E.evaluate_scores(fs, v["user_scores"], metric="pearson")
    with pytest.raises(KeyError):
        E.generate_summary(sc, v["change_points"], v["n_frames"], v["n_frame_per_seg"].tolist(), v["picks"], 0.15, "greedy")


def test_upsample_edge_cases_vs_literal_oracle():
    rng = np.random.default_rng(0)
    for T, n_frames, picks in [
        (5, 75, 15 * np.arange(5)),                     # sentinel appended
        (5, 60, np.array([0, 15, 30, 45, 60])),         # last pick == n_frames: no sentinel, 4 intervals
        (3, 50, np.array([5, 20, 35])),                 # first pick > 0 -> leading zeros
        (1, 10, np.array([0])),
        (4, 61, np.array([0, 15, 30, 45], dtype=np.float32)),   # non-int dtype is cast (eval.py:25-26)
    ]:
        sc = rng.random(T).astype(np.float32)
        np.testing.assert_array_equal(E.upsample(sc, n_frames, picks), eval_np.upsample(sc, n_frames, picks))


def test_knapsack_native_matches_oracle_and_bruteforce_value():
    rng = np.random.default_rng(11)
    for trial in range(200):
        n = int(rng.integers(1, 14))
        vals = rng.random(n).tolist()
        wts = rng.integers(1, 60, n).tolist()
        cap = int(rng.integers(0, 200))
        picks = knapsack_ortools(vals, wts, n, cap)
        assert picks == knapsack_np.knapsack_dp(vals, wts, n, cap)          # bit-exact vs the CPU restatement
        v, w = knapsack_np.knapsack_value(vals, wts, picks)
        assert w <= cap
        if trial < 60:
            assert v == knapsack_np.knapsack_bruteforce_value(vals, wts, cap)


def test_knapsack_summary_budget_and_equality_with_oracle():
    for ci, T in enumerate([300, 120, 640]):
        v = R.synthetic_video(T, 40 + ci, n_users=5)
        sc = np.random.default_rng(ci).random(T).astype(np.float32)
        args = (sc, v["change_points"], v["n_frames"], v["n_frame_per_seg"].tolist(), v["picks"], 0.15, "knapsack")
        a = E.generate_summary(*args)
        b = eval_np.generate_summary(*args)
        np.testing.assert_array_equal(a, b)
        assert a.sum() <= int(np.floor(v["n_frames"] * 0.15))
        assert a.shape == (v["n_frames"],)


def test_native_eval_tail_is_bit_identical_to_numpy():
    """sumk_eval_videos vs utils/eval.py: summaries and F-scores bit for bit (incl. the float64 promotion when the summary is
    shorter than the video, and trimming when longer), Spearman to 1e-12, for both selection methods."""
    from summarizer_amd.utils import eval_native
    vids, scores, refs = [], [], []
    for ci, (T, U, tweak) in enumerate([(300, 15, 0), (120, 18, 0), (17, 3, 0), (640, 20, 0), (200, 7, -9), (150, 5, +6), (90, 4, 0)]):
        v = R.synthetic_video(T, 300 + ci, n_users=U)
        nfps = v["n_frame_per_seg"].copy()
        nfps[-1] += tweak                                     # sum(nfps) != n_frames: padding / trimming branches
        sc = np.random.default_rng(400 + ci).random(T).astype(np.float32)
        if ci == 6:
            sc[:] = np.float32(0.25)                          # all ties: average ranks, constant machine vector -> NaN corr
        ranks = E.rank_users(v["user_scores"])
        vids.append(eval_native.prepare_video(v["n_frames"], v["picks"], v["change_points"], nfps, v["user_summary"], ranks))
        scores.append(sc)
        refs.append((v, nfps, sc))
    for method in ("knapsack", "rank"):
        corr, f_avg, f_max, summ = eval_native.evaluate_batch(vids, scores, 0.15, method, want_summaries=True, n_threads=3)
        for i, (v, nfps, sc) in enumerate(refs):
            s_ref = E.generate_summary(sc, v["change_points"], v["n_frames"], nfps.tolist(), v["picks"], 0.15, method)
            np.testing.assert_array_equal(summ[i], s_ref)
            fa, fm = E.evaluate_summary(s_ref, v["user_summary"])
            assert float(fa) == f_avg[i] and float(fm) == f_max[i], (method, i, fa, f_avg[i], fm, f_max[i])
            with np.errstate(invalid="ignore"):
                c_ref = E.evaluate_scores(E.upsample(sc, v["n_frames"], v["picks"]), v["user_scores"])
            np.testing.assert_allclose(corr[i], c_ref, rtol=1e-12, atol=1e-14, equal_nan=True)
    with pytest.raises(KeyError):
        eval_native.evaluate_batch(vids, scores, 0.15, "greedy")
    # default thread count = the library's persistent worker pool: same numbers as three fresh threads, call after call, and in a child
    # process forked AFTER the pool exists (its workers do not survive the fork: the child builds its own)
    want = eval_native.evaluate_batch(vids, scores, 0.15, "knapsack", n_threads=3)
    for _ in range(3):
        got = eval_native.evaluate_batch(vids, scores, 0.15, "knapsack")
        for a, b in zip(got[:3], want[:3]):
            np.testing.assert_array_equal(a, b)
    import multiprocessing as mp
    q = mp.get_context("fork").Queue()
    def child():
        g = eval_native.evaluate_batch(vids, scores, 0.15, "knapsack")
        q.put([np.asarray(x).tolist() for x in g[:3]])
    pr = mp.get_context("fork").Process(target=child)
    pr.start()
    res = q.get(timeout=120)
    pr.join(timeout=60)
    assert pr.exitcode == 0
    for a, b in zip(res, want[:3]):
        np.testing.assert_array_equal(np.asarray(a), b)


def test_pack_rows_native_threads():
    """sumk_pack_rows (host): ragged videos packed back to back, any thread count, chunk seams inside and across videos."""
    import ctypes as C
    from summarizer_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    D = 24
    lens = [1, 7, 300, 2, 4500, 64, 1, 9000]
    arrs = [rng.standard_normal((T, D)).astype(np.float32) for T in lens]
    want = np.concatenate(arrs)
    for nt in (0, 1, 2, 5, 16):
        dst = np.full_like(want, np.nan)
        srcs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        rc = lib.sumk_pack_rows(C.c_void_p(dst.ctypes.data), srcs, _lib.host_i32(np.asarray(lens, np.int32)), len(arrs), D, nt)
        assert rc == 0
        np.testing.assert_array_equal(dst, want)
    assert lib.sumk_pack_rows(None, None, None, 0, D, 0) == 0
    bad = (C.c_void_p * 1)(None)
    assert lib.sumk_pack_rows(C.c_void_p(dst.ctypes.data), bad, _lib.host_i32(np.asarray([3], np.int32)), 1, D, 1) != 0


def test_pack_rows_bf16_native_matches_round_to_nearest_even():
    """sumk_pack_rows_bf16 (host): the same packing with fp32 -> bf16 folded into the copy; bit patterns = torch's round-to-nearest-even
    conversion, ties, subnormals, infinities and NaNs (which must stay NaNs) included; any thread count."""
    import ctypes as C
    import torch
    from summarizer_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(4)
    D = 16
    lens = [1, 5, 700, 3, 4100]
    arrs = [rng.standard_normal((T, D)).astype(np.float32) * np.float32(10.0 ** rng.integers(-30, 30)) for T in lens]
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-40, 3.4e38, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -9,
                        np.float32(1.00390625), np.float32(1.01171875), 65504.0, -2.5e-39, 7.0, 1e30], dtype=np.float32)   # incl. exact ties
    arrs[1][0, :] = special
    want = torch.from_numpy(np.concatenate(arrs)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    for nt in (0, 1, 3, 16):
        dst = np.zeros(want.shape, dtype=np.uint16)
        srcs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        assert lib.sumk_pack_rows_bf16(C.c_void_p(dst.ctypes.data), srcs, _lib.host_i32(np.asarray(lens, np.int32)), len(arrs), D, nt) == 0
        nan = np.isnan(np.concatenate(arrs))
        np.testing.assert_array_equal(dst[~nan], want[~nan])
        assert ((dst[nan] & 0x7F80) == 0x7F80).all() and ((dst[nan] & 0x007F) != 0).all()        # still NaNs
    assert lib.sumk_pack_rows_bf16(None, None, None, 0, D, 0) == 0


# ------------------------------------------------------------------------------------------------ knapsack: the tie-free surface is pinned
def _all_optimal_subsets(v, w, cap):
    n = len(v)
    sv = np.zeros(1, dtype=np.int64); sw = np.zeros(1, dtype=np.int64)
    for i in range(n):
        sv = np.concatenate([sv, sv + v[i]]); sw = np.concatenate([sw, sw + w[i]])
    feas = sw <= cap
    best = sv[feas].max()
    return [[i for i in range(n) if (m >> i) & 1] for m in np.flatnonzero(feas & (sv == best))]


def test_knapsack_selected_set_equals_the_unique_optimum():
    """Wherever the optimal subset is UNIQUE every exact solver (OR-tools' DP included) returns the same set: on such
    instances the native DP and the CPU restatement must return exactly that set, not merely its value (S <= 18, exhaustive)."""
    rng = np.random.default_rng(23)
    unique = tied = 0
    for trial in range(400):
        n = int(rng.integers(1, 19))
        style = trial % 4
        vals = rng.random(n)
        if style == 1:
            vals = np.round(vals, 2)                                  # coarse values: many ties (those instances are skipped)
        elif style == 2:
            vals = 0.5 + 0.01 * vals
        wts = rng.integers(1, 400, n)
        cap = int(rng.integers(0, max(2, int(wts.sum() * rng.uniform(0.1, 0.9)))))
        v, w = knapsack_np.to_int_problem(vals.tolist(), wts.tolist())
        opt = _all_optimal_subsets(v, w, cap)
        if len(opt) != 1:
            tied += 1
            continue
        unique += 1
        assert knapsack_ortools(vals.tolist(), wts.tolist(), n, cap) == opt[0], (trial, vals, wts, cap)
        assert knapsack_np.knapsack_dp(vals.tolist(), wts.tolist(), n, cap) == opt[0]
    assert unique >= 200 and tied >= 20, (unique, tied)


def test_knapsack_e2e_vs_reference_goldens():
    """tests/golden/knapsack_e2e.npz: the REAL reference's generate_summary(method="knapsack") + evaluate_summary with only
    OR-tools' solver swapped for an exhaustive one, unique-optimum videos only (make_golden_knapsack.py).  The numpy mirror,
    the native threaded tail and the oracle must reproduce every machine summary bit for bit and both F-scores."""
    from summarizer_amd.utils import eval_native
    g = load_golden("knapsack_e2e")
    n = int(g["n_cases"])
    assert n >= 20
    vids, scores = [], []
    for ci in range(n):
        T, U, seed = (int(x) for x in g[f"c{ci}/meta"])
        v = R.synthetic_video(T, seed, n_users=U)
        sc = g[f"c{ci}/scores"]
        args = (sc, v["change_points"], v["n_frames"], v["n_frame_per_seg"].tolist(), v["picks"], 0.15, "knapsack")
        summ = E.generate_summary(*args)
        np.testing.assert_array_equal(summ, g[f"c{ci}/summary"].astype(np.float32), err_msg=f"case {ci}")
        np.testing.assert_array_equal(eval_np.generate_summary(*args), summ)
        np.testing.assert_array_equal(np.array(E.evaluate_summary(summ, v["user_summary"])), g[f"c{ci}/fscore"])
        vids.append(eval_native.prepare_video(v["n_frames"], v["picks"], v["change_points"], v["n_frame_per_seg"], v["user_summary"],
                                              E.rank_users(v["user_scores"])))
        scores.append(sc)
    _, f_avg, f_max, summ = eval_native.evaluate_batch(vids, scores, 0.15, "knapsack", want_summaries=True, n_threads=4)
    for ci in range(n):
        np.testing.assert_array_equal(summ[ci], g[f"c{ci}/summary"].astype(np.float32))
        assert (f_avg[ci], f_max[ci]) == tuple(g[f"c{ci}/fscore"]), ci


def test_trainer_evaluation_with_knapsack_selection_vs_reference_goldens():
    """`Trainer.test`'s evaluation tail with hps.selection_algorithm="knapsack" (the reference default, config.py:36) on the
    golden videos: fold-level mean F-scores equal the means of the reference's per-video values."""
    from summarizer_amd.models import Trainer
    from summarizer_amd.utils.datasets import DictDataset
    from summarizer_amd.utils.hps import make_hps
    g = load_golden("knapsack_e2e")
    n = int(g["n_cases"])
    videos, acts = {}, {}
    for ci in range(n):
        T, U, seed = (int(x) for x in g[f"c{ci}/meta"])
        v = R.synthetic_video(T, seed, n_users=U)
        videos[f"video_{ci+1}"] = v
        acts[f"video_{ci+1}"] = g[f"c{ci}/scores"]
    keys = list(videos)
    hps = make_hps(DictDataset(videos), [{"train_keys": [], "test_keys": keys}], selection_algorithm="knapsack")
    tr = Trainer(hps, hps.splits_files[0])
    _, f_avg, f_max, _ = tr._evaluate_native(acts, keys)
    ref = np.array([g[f"c{ci}/fscore"] for ci in range(n)])
    assert np.mean(f_avg) == np.mean(ref[:, 0]) and np.mean(f_max) == np.mean(ref[:, 1])
    np.testing.assert_allclose(tr._eval_summary(acts, keys), (np.mean(ref[:, 0]), np.mean(ref[:, 1])), rtol=1e-6)   # per-video numpy path (float32 means)
