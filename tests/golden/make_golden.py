#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REAL reference (/root/reference, read-only, imported with
in-memory stubs for the packages it needs but the image lacks: h5py, ortools).  Run ONCE in the build
container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
The reference never travels to the GPU box; only the small .npz vectors (inputs + expected outputs) do.
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

import torch
import recipes as R
from summarizer.models.vasnet import VASNet
from summarizer.models.dsn import DSN, DSNTrainer
from summarizer.models.sumgan import sLSTM
from summarizer.utils import eval as ref_eval

torch.set_num_threads(4)


def t(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KB")


# ----------------------------------------------------------------------------- G1/G3 VASNet small-D
VARIANTS = {
    "default": dict(),
    "local5": dict(attention_aperture=5),
    "ignore_self": dict(ignore_self=True),
    "scale006": dict(scale=0.06),
    "local3_ignore": dict(attention_aperture=3, ignore_self=True),
    "pos_simple": dict(max_length=64, pos_embed="simple"),
    "pos_attention": dict(max_length=64, pos_embed="attention"),
    "eps1e-3": dict(epsilon=1e-3),
}
D = 64
out = {}
meta = {}
for vi, (vname, kw) in enumerate(VARIANTS.items()):
    m = VASNet(input_size=D, **kw).eval()
    w = R.vasnet_weights(D, 100 + vi, max_length=kw.get("max_length") if kw.get("pos_embed") == "simple" else None)
    m.load_state_dict(t(w))
    for k, v in w.items():
        out[f"{vname}/w/{k}"] = v
    if kw.get("pos_embed") == "attention":
        out[f"{vname}/pos_table"] = m.pos_embed.numpy().copy()
    Ts = [1, 2, 37] if kw.get("max_length") else [1, 2, 37, 130]
    for T in Ts:
        for B in ([1, 3] if T == 37 else [1]):
            x = R.features(T, B, D, 1000 * vi + T + B) - 0.2   # mixed sign so logits span both signs
            out[f"{vname}/x/T{T}B{B}"] = x
            with torch.no_grad():
                y = m(torch.from_numpy(x.copy()))
            out[f"{vname}/y/T{T}B{B}"] = y.numpy()
    meta[vname] = {k: (v if not isinstance(v, float) else float(v)) for k, v in kw.items()}
out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
save("vasnet_small", **out)

# G3: intermediates for one T=37 default case (hooks on the real module)
m = VASNet(input_size=D).eval(); w = R.vasnet_weights(D, 100); m.load_state_dict(t(w))
x = torch.from_numpy(R.features(37, 1, D, 1000 * 0 + 37 + 1) - 0.2)
cap = {}
hooks = [getattr(m, n).register_forward_hook(lambda mod, i, o, n=n: cap.setdefault(n, []).append(o.detach().numpy().copy()))
         for n in ["K", "Q", "V", "softmax", "attention_head_projection", "layer_norm", "k1", "k2"]]
with torch.no_grad():
    y = m(x.clone())
for h in hooks: h.remove()
save("vasnet_intermediates", x=x.numpy(), y=y.numpy(), Q=cap["Q"][0], K=cap["K"][0], V=cap["V"][0],
     alpha=cap["softmax"][0], c=cap["attention_head_projection"][0], y1=cap["layer_norm"][0],
     k1=cap["k1"][0], y2=cap["layer_norm"][1], k2=cap["k2"][0])

# ----------------------------------------------------------------------------- G2 VASNet full-D (outputs only)
out = {}
for ci, (kw, T, B) in enumerate([(dict(), 300, 1), (dict(), 163, 1), (dict(attention_aperture=16), 320, 1),
                                 (dict(ignore_self=True), 97, 2), (dict(), 1, 1), (dict(), 640, 1)]):
    Dm = 1024
    m = VASNet(input_size=Dm, **kw).eval()
    w = R.vasnet_weights(Dm, 7000 + ci); m.load_state_dict(t(w))
    x = R.features(T, B, Dm, 7100 + ci)
    with torch.no_grad():
        y = m(torch.from_numpy(x.copy()))
    out[f"c{ci}/y"] = y.numpy()
    out[f"c{ci}/cfg"] = np.frombuffer(json.dumps(dict(kw=kw, T=T, B=B, D=Dm, wseed=7000 + ci, xseed=7100 + ci,
                                                     wdigest=R.digest(w), xdigest=R.digest({"x": x}))).encode(), dtype=np.uint8)
save("vasnet_full", **out)

# ----------------------------------------------------------------------------- G1/G2 LSTM scorers
out = {}
cases = [("dsn_small", "dsn", 64, 16, 1, [1, 2, 37, 130]), ("dsn_small_2l", "dsn", 64, 16, 2, [37]),
         ("slstm_small", "slstm", 64, 32, 2, [1, 37, 90])]
for ci, (name, kind, Dm, H, L, Ts) in enumerate(cases):
    if kind == "dsn":
        m = DSN(input_size=Dm, hidden_size=H, num_layers=L).eval(); w = R.lstm_weights("rnn.", Dm, H, L, 300 + ci, "out.0.")
    else:
        m = sLSTM(input_size=Dm, hidden_size=H, num_layers=L).eval(); w = R.lstm_weights("lstm.", Dm, H, L, 300 + ci, "out.")
    m.load_state_dict(t(w))
    for k, v in w.items(): out[f"{name}/w/{k}"] = v
    for T in Ts:
        for B in ([1, 3] if T == 37 else [1]):
            x = R.features(T, B, Dm, 3000 + 10 * ci + T + B) - 0.2
            out[f"{name}/x/T{T}B{B}"] = x
            with torch.no_grad():
                y = m(torch.from_numpy(x.copy()))
            out[f"{name}/y/T{T}B{B}"] = y.numpy()
            if T == 37 and B == 1:
                with torch.no_grad():
                    h, _ = (m.rnn if kind == "dsn" else m.lstm)(torch.from_numpy(x.copy()))
                out[f"{name}/h/T{T}B{B}"] = h.numpy()
save("lstm_small", **out)

out = {}
for ci, (kind, Dm, H, L, T) in enumerate([("dsn", 1024, 256, 1, 300), ("dsn", 1024, 256, 1, 41), ("slstm", 1024, 1024, 2, 60)]):
    if kind == "dsn":
        m = DSN(input_size=Dm, hidden_size=H, num_layers=L).eval(); w = R.lstm_weights("rnn.", Dm, H, L, 8000 + ci, "out.0.")
    else:
        m = sLSTM(input_size=Dm, hidden_size=H, num_layers=L).eval(); w = R.lstm_weights("lstm.", Dm, H, L, 8000 + ci, "out.")
    m.load_state_dict(t(w))
    x = R.features(T, 1, Dm, 8100 + ci)
    with torch.no_grad():
        y = m(torch.from_numpy(x.copy()))
    out[f"c{ci}/y"] = y.numpy()
    out[f"c{ci}/cfg"] = np.frombuffer(json.dumps(dict(kind=kind, D=Dm, H=H, L=L, T=T, wseed=8000 + ci, xseed=8100 + ci,
                                                     wdigest=R.digest(w))).encode(), dtype=np.uint8)
save("lstm_full", **out)

# ----------------------------------------------------------------------------- G4 training steps (dropout off: eval() + grads)
def train_golden(m, w, x, target, steps, loss_fn):
    m.load_state_dict(t(w)); m.eval()
    opt = torch.optim.Adam(m.parameters(), lr=5e-5, weight_decay=1e-5)    # vasnet.py:181, config.py:28-29
    rec = {}
    for s in range(steps):
        scores = m(torch.from_numpy(x.copy()))
        loss = loss_fn(scores, torch.from_numpy(target))
        opt.zero_grad(); loss.backward()
        if s == 0:
            rec["loss0"] = np.float32(loss.item())
            for k, p_ in m.named_parameters(): rec[f"grad0/{k}"] = p_.grad.numpy().copy()
        opt.step()
        if s in (0, steps - 1):
            for k, p_ in m.named_parameters(): rec[f"param{s+1}/{k}"] = p_.detach().numpy().copy()
        rec[f"loss{s}"] = np.float32(loss.item())
    return rec

out = {}
T = 37
tgt = np.random.default_rng(5).random((T, 1, 1)).astype(np.float32)
x = R.features(T, 1, 64, 4242) - 0.2
w = R.vasnet_weights(64, 4000)
rec = train_golden(VASNet(input_size=64), w, x, tgt, 3, torch.nn.MSELoss())
out.update({f"vasnet/{k}": v for k, v in rec.items()}); out["vasnet/x"] = x; out["vasnet/target"] = tgt
for k, v in w.items(): out[f"vasnet/w/{k}"] = v
w = R.vasnet_weights(64, 4001)
rec = train_golden(VASNet(input_size=64, attention_aperture=4, ignore_self=True), w, x, tgt, 3, torch.nn.MSELoss())
out.update({f"vasnet_loc/{k}": v for k, v in rec.items()})
for k, v in w.items(): out[f"vasnet_loc/w/{k}"] = v
w = R.lstm_weights("rnn.", 64, 16, 1, 4002, "out.0.")
rec = train_golden(DSN(input_size=64, hidden_size=16), w, x, tgt, 3, torch.nn.MSELoss())
out.update({f"dsn/{k}": v for k, v in rec.items()})
for k, v in w.items(): out[f"dsn/w/{k}"] = v
rec = train_golden(DSN(input_size=64, hidden_size=16), w, x, tgt, 1, torch.nn.BCELoss())     # dsn.py:76,117-119
out.update({f"dsn_bce/{k}": v for k, v in rec.items()})
save("train_small", **out)

# ----------------------------------------------------------------------------- G5 reward (dsn.py:185-236)
tr = DSNTrainer.__new__(DSNTrainer)
tr.hps = types.SimpleNamespace(use_cuda=False)
out = {}
rng = np.random.default_rng(77)
for ci, (T, Dm, pr) in enumerate([(60, 64, 0.5), (60, 64, 0.1), (200, 128, 0.5), (37, 64, 0.0), (37, 64, -1)]):
    seq = R.features(T, 1, Dm, 500 + ci)
    if pr == -1:
        act = np.zeros((T, 1, 1), np.float32); act[11] = 1       # exactly one pick
    else:
        act = (rng.random((T, 1, 1)) < pr).astype(np.float32)    # pr=0 -> zero picks
    for far in (False, True):
        try:
            r = tr.compute_reward(torch.from_numpy(seq), torch.from_numpy(act), far_sim=far, temp_dist_thre=20)
            out[f"c{ci}/reward_far{int(far)}"] = np.float32(r.item())
        except IndexError:
            # exactly ONE pick: the reference indexes with a 0-dim tensor (dsn.py:229) and .min(1) raises.
            out[f"c{ci}/reward_far{int(far)}"] = np.float32(np.nan)
            out[f"c{ci}/raises_IndexError"] = np.int32(1)
    out[f"c{ci}/seq"] = seq; out[f"c{ci}/actions"] = act
save("reward", **out)

# ----------------------------------------------------------------------------- G6 metrics (eval.py)
out = {}
for ci, (T, U) in enumerate([(300, 15), (120, 18), (17, 3)]):
    v = R.synthetic_video(T, 900 + ci, n_users=U)
    scores = np.random.default_rng(950 + ci).random(T).astype(np.float32)
    fs = ref_eval.upsample(scores, v["n_frames"], v["picks"])
    summ = ref_eval.generate_summary(scores, v["change_points"], v["n_frames"], v["n_frame_per_seg"].tolist(), v["picks"], 0.15, "rank")
    f_avg, f_max = ref_eval.evaluate_summary(summ, v["user_summary"])
    corr = ref_eval.evaluate_scores(fs, v["user_scores"], metric="spearmanr")
    ken = ref_eval.evaluate_scores(fs, v["user_scores"], metric="kendalltau") if T <= 120 else np.nan
    out[f"c{ci}/scores"] = scores; out[f"c{ci}/frame_scores"] = fs; out[f"c{ci}/summary_rank"] = summ
    out[f"c{ci}/fscore"] = np.array([f_avg, f_max], dtype=np.float64); out[f"c{ci}/spearman"] = np.float64(corr)
    out[f"c{ci}/kendall"] = np.float64(ken)
    out[f"c{ci}/T_U_seed"] = np.array([T, U, 900 + ci])
    # short machine summary (padding branch eval.py:141-143) and long one (trim branch :139-140)
    out[f"c{ci}/fscore_short"] = np.array(ref_eval.evaluate_summary(summ[:-7], v["user_summary"]), dtype=np.float64)
    out[f"c{ci}/fscore_long"] = np.array(ref_eval.evaluate_summary(np.concatenate([summ, np.ones(5, np.float32)]), v["user_summary"]), dtype=np.float64)
save("metrics", **out)
print("done")
