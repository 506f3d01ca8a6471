"""CPU: pins the oracle (numpy restatements + torch port) against vectors produced by the REAL reference
(tests/golden/make_golden.py).  Tolerances: fp32 restatement vs fp32 reference -> 2e-5 abs on scores in (0,1)."""
import numpy as np
import torch

import recipes as R
from conftest import load_golden, js
from oracle import vasnet_np, lstm_np, reward_np, eval_np, knapsack_np, torch_port

TOL = 2e-5


def _kw(meta):
    return dict(ignore_self=meta.get("ignore_self", False), aperture=meta.get("attention_aperture"),
                scale=meta.get("scale"), eps=meta.get("epsilon", 1e-6))


def _cases(g, vname):
    return sorted(k.split("/")[-1] for k in g.files if k.startswith(f"{vname}/x/"))


def test_vasnet_small_numpy_and_torch_port():
    g = load_golden("vasnet_small")
    meta = js(g["meta"])
    for vname, m in meta.items():
        w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith(f"{vname}/w/")}
        pos = None
        if m.get("max_length"):
            pos = w.pop("pos_embed.weight") if m["pos_embed"] == "simple" else g[f"{vname}/pos_table"]
            if m["pos_embed"] == "attention":
                np.testing.assert_allclose(vasnet_np.sinusoid_table(m["max_length"], 64), pos, atol=1e-6)
        for c in _cases(g, vname):
            x, y = g[f"{vname}/x/{c}"], g[f"{vname}/y/{c}"]
            for dt in (np.float32, np.float64):
                yo = vasnet_np.vasnet_forward(x, w, pos_table=pos, pos_kind=m.get("pos_embed", "simple"), dtype=dt, **_kw(m))
                assert yo.shape == y.shape
                np.testing.assert_allclose(yo, y, atol=TOL, rtol=0, equal_nan=True, err_msg=f"{vname} {c} {dt}")
            yt = torch_port.vasnet_scores(torch.from_numpy(x), {k: torch.from_numpy(v) for k, v in w.items()},
                                          pos_table=None if pos is None else torch.from_numpy(pos),
                                          pos_kind=m.get("pos_embed", "simple"), **_kw(m)).numpy()
            np.testing.assert_allclose(yt, y, atol=TOL, rtol=0, equal_nan=True, err_msg=f"torch {vname} {c}")


def test_vasnet_ignore_self_T1_is_nan():
    # all keys masked -> softmax of all -inf -> NaN in the reference; the oracle must agree, not "fix" it
    g = load_golden("vasnet_small")
    assert np.isnan(g["ignore_self/y/T1B1"]).all()


def test_vasnet_intermediates():
    g = load_golden("vasnet_intermediates")
    w = R.vasnet_weights(64, 100)
    y, it = vasnet_np.vasnet_forward(g["x"], w, return_intermediates=True)
    np.testing.assert_allclose(y, g["y"], atol=TOL)
    for a, b in [("Q", "Q"), ("K", "K"), ("V", "V"), ("alpha", "alpha"), ("c", "c"), ("y1", "y1"), ("y2", "y2")]:
        np.testing.assert_allclose(it[a], g[b][0], atol=5e-5, rtol=1e-4, err_msg=a)


def test_vasnet_full_size():
    g = load_golden("vasnet_full")
    n = len([k for k in g.files if k.endswith("/cfg")])
    for ci in range(n):
        cfg = js(g[f"c{ci}/cfg"])
        if cfg["T"] > 320:
            continue
        w = R.vasnet_weights(cfg["D"], cfg["wseed"]); x = R.features(cfg["T"], cfg["B"], cfg["D"], cfg["xseed"])
        assert R.digest(w) == cfg["wdigest"] and R.digest({"x": x}) == cfg["xdigest"]
        kw = cfg["kw"]
        y = vasnet_np.vasnet_forward(x, w, ignore_self=kw.get("ignore_self", False), aperture=kw.get("attention_aperture"))
        np.testing.assert_allclose(y, g[f"c{ci}/y"], atol=TOL, err_msg=str(cfg))


def test_lstm_small():
    g = load_golden("lstm_small")
    for name, fn, L in [("dsn_small", lstm_np.dsn_forward, 1), ("dsn_small_2l", lstm_np.dsn_forward, 2),
                        ("slstm_small", lstm_np.slstm_forward, 2)]:
        w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith(f"{name}/w/")}
        for c in _cases(g, name):
            y = fn(g[f"{name}/x/{c}"], w, num_layers=L)
            np.testing.assert_allclose(y, g[f"{name}/y/{c}"], atol=TOL, err_msg=f"{name} {c}")
        pref = "rnn." if name.startswith("dsn") else "lstm."
        h = lstm_np.bilstm_forward(g[f"{name}/x/T37B1"], w, pref, L)
        np.testing.assert_allclose(h, g[f"{name}/h/T37B1"], atol=TOL)


def test_lstm_full_dsn():
    g = load_golden("lstm_full")
    cfg = js(g["c1/cfg"])
    w = R.lstm_weights("rnn.", cfg["D"], cfg["H"], cfg["L"], cfg["wseed"], "out.0.")
    assert R.digest(w) == cfg["wdigest"]
    y = lstm_np.dsn_forward(R.features(cfg["T"], 1, cfg["D"], cfg["xseed"]), w)
    np.testing.assert_allclose(y, g["c1/y"], atol=TOL)


def test_reward():
    g = load_golden("reward")
    for ci in range(5):
        for far in (0, 1):
            ref = g[f"c{ci}/reward_far{far}"]
            got = reward_np.compute_reward(g[f"c{ci}/seq"], g[f"c{ci}/actions"], far_sim=bool(far))
            if f"c{ci}/raises_IndexError" in g.files:
                # one pick: the reference crashes (dsn.py:229-230); the oracle defines div=0, rep=exp(-mean d)
                assert np.isfinite(got) and 0 < got <= 0.5
            else:
                np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)
    assert reward_np.compute_reward(g["c3/seq"], g["c3/actions"]) == 0.0      # zero picks (dsn.py:199-203)


def test_metrics():
    g = load_golden("metrics")
    for ci in range(3):
        T, U, seed = g[f"c{ci}/T_U_seed"]
        v = R.synthetic_video(int(T), int(seed), n_users=int(U))
        sc = g[f"c{ci}/scores"]
        fs = eval_np.upsample(sc, v["n_frames"], v["picks"])
        np.testing.assert_array_equal(fs, g[f"c{ci}/frame_scores"])
        summ = eval_np.generate_summary(sc, v["change_points"], v["n_frames"], v["n_frame_per_seg"].tolist(), v["picks"], 0.15, "rank")
        np.testing.assert_array_equal(summ, g[f"c{ci}/summary_rank"])
        np.testing.assert_allclose(eval_np.evaluate_summary(summ, v["user_summary"]), g[f"c{ci}/fscore"], rtol=1e-12)
        np.testing.assert_allclose(eval_np.evaluate_summary(summ[:-7], v["user_summary"]), g[f"c{ci}/fscore_short"], rtol=1e-12)
        long = np.concatenate([summ, np.ones(5, np.float32)])
        np.testing.assert_allclose(eval_np.evaluate_summary(long, v["user_summary"]), g[f"c{ci}/fscore_long"], rtol=1e-12)
        np.testing.assert_allclose(eval_np.evaluate_scores(fs, v["user_scores"]), g[f"c{ci}/spearman"], rtol=1e-12)


def test_knapsack_value_optimal_vs_bruteforce():
    rng = np.random.default_rng(3)
    for trial in range(60):
        n = int(rng.integers(1, 13))
        vals = rng.random(n).tolist()
        wts = rng.integers(1, 40, n).tolist()
        cap = int(rng.integers(0, 120))
        picks = knapsack_np.knapsack_dp(vals, wts, n, cap)
        v, w = knapsack_np.knapsack_value(vals, wts, picks)
        assert w <= cap
        assert v == knapsack_np.knapsack_bruteforce_value(vals, wts, cap), (vals, wts, cap, picks)


def test_transformer_oracle_vs_reference_goldens():
    from oracle import transformer_np
    g = load_golden("transformer_small")
    meta = js(g["meta"])
    for name, kw in meta.items():
        w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith(f"{name}/w/")}
        pos = w.get("pos_embed.weight") if kw.get("max_length") else None
        for c in _cases(g, name):
            y = transformer_np.transformer_forward(g[f"{name}/x/{c}"], w, kw["encoder_layers"], kw["attention_heads"],
                                                   eps=kw.get("epsilon", 1e-5), more_residuals=kw.get("more_residuals", False),
                                                   pos_table=pos)
            np.testing.assert_allclose(y, g[f"{name}/y/{c}"], atol=TOL, rtol=0, err_msg=f"{name} {c}")
    g = load_golden("transformer_full")
    cfg = js(g["c1/cfg"])
    w = R.transformer_weights(cfg["D"], cfg["layers"], cfg["wseed"])
    assert R.digest(w) == cfg["wdigest"]
    y = transformer_np.transformer_forward(R.features(cfg["T"], cfg["B"], cfg["D"], cfg["xseed"]), w, cfg["layers"], cfg["heads"])
    np.testing.assert_allclose(y, g["c1/y"], atol=TOL, rtol=0)


def test_c1_summe_fold0_oracle_vs_the_reference_at_size():
    """BASELINE config 1 at size: the oracle (torch port scores, numpy evaluation tail, knapsack restatement) against the REAL
    reference's Trainer.test(fold 0) on S-SumMe (tests/golden/make_golden_c1.py): scores 2e-5, machine summaries bit-exact under both
    selection algorithms, per-video F-scores and Spearman, and the fold means Trainer.test returned."""
    import hashlib, json
    from summarizer_amd.models.vasnet import VASNet
    from summarizer_amd.utils.datasets import synthetic_dataset
    g = load_golden("c1_summe_fold0")
    meta = json.loads(bytes(g["meta"]).decode())
    ds = synthetic_dataset(meta["n_videos"], seed=meta["dataset_seed"], D=meta["D"], t_range=tuple(meta["t_range"]), n_users=meta["n_users"])
    torch.manual_seed(meta["weight_seed"])
    p = {k: v.detach() for k, v in VASNet(input_size=meta["D"]).named_parameters()}
    for k, v in p.items():
        assert hashlib.sha256(np.ascontiguousarray(v.numpy()).tobytes()).digest() == bytes(g[f"wdigest/{k}"]), k
    keys = meta["fold0"]["test_keys"]
    corrs, fs = [], {"rank": [], "knapsack": []}
    torch.set_num_threads(4)
    for k in keys:
        d = ds[k]
        assert hashlib.sha256(np.ascontiguousarray(d["features"][...]).tobytes()).digest() == bytes(g[f"digest/{k}"]), k
        with torch.no_grad():
            y = torch_port.vasnet_scores(torch.from_numpy(d["features"][...]).unsqueeze(1), p)[:, 0, 0].numpy()
        np.testing.assert_allclose(y, g[f"scores/{k}"], atol=TOL, rtol=0)
        n_frames = int(d["n_frames"][()])
        for algo in ("rank", "knapsack"):
            summ = eval_np.generate_summary(y, d["change_points"][...], n_frames, d["n_frame_per_seg"][...].tolist(), d["picks"][...], 0.15, algo)
            np.testing.assert_array_equal(summ, np.unpackbits(g[f"summary_{algo}/{k}"])[:n_frames].astype(np.float32), err_msg=f"{algo} {k}")
            f = eval_np.evaluate_summary(summ, d["user_summary"][...])
            np.testing.assert_allclose(f, g[f"fscore_{algo}/{k}"], rtol=1e-6)
            fs[algo].append(f)
        c = eval_np.evaluate_scores(eval_np.upsample(y, n_frames, d["picks"][...]), d["user_scores"][...])
        np.testing.assert_allclose(c, float(g[f"corr/{k}"]), atol=5e-5)
        corrs.append(c)
    for algo in ("rank", "knapsack"):
        ref = g[f"test_{algo}"]
        np.testing.assert_allclose(np.mean(corrs), ref[0], atol=2e-5)
        np.testing.assert_allclose(np.mean(np.array(fs[algo]), axis=0), ref[1:], rtol=1e-6)
