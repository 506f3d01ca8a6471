#!/usr/bin/env python3
"""Full-size TRAINING goldens from the REAL reference (/root/reference, read-only; in-memory stubs for h5py / ortools):
the BASELINE shapes (D = 1024; VASNet T = 300 and a 3-video accumulated batch, DSN H = 256, sLSTM H = 1024 x 2 layers) run
through the reference modules exactly as the reference trainers do one step (vasnet.py:196-212, dsn.py:104-146: MSELoss on
the scores, backward, Adam(lr 5e-5, weight_decay 1e-5)), with dropout off (`.eval()`, like train_small.npz).

Weights and inputs come from seeded recipes (tests/golden/recipes.py), so only DIGESTS of the 21-170 MB gradients are
committed: per parameter the L2 norm, the absolute maximum and 256 sampled entries (flat indices drawn from a seeded
generator), for the gradient and for the parameter after one Adam step.  Run once in the build container:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train_full.py
"""
import os, sys, types
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
from summarizer.models.dsn import DSN
from summarizer.models.sumgan import sLSTM

torch.set_num_threads(8)
N_SAMPLE = 256


def sample_idx(name, numel):
    """Flat indices sampled per parameter; the tests call the same function (recipes.sample_idx)."""
    return R.sample_idx(name, numel, N_SAMPLE)


def digest_into(out, tag, named):
    for k, v in named:
        a = v.detach().numpy().reshape(-1).astype(np.float64)
        out[f"{tag}/{k}/norm"] = np.float64(np.sqrt((a * a).sum()))
        out[f"{tag}/{k}/absmax"] = np.float64(np.abs(a).max())
        out[f"{tag}/{k}/sample"] = a[sample_idx(k, a.size)].astype(np.float32)


def one_step(model, videos, out, tag):
    """videos: list of (x (T,1,D), target (T,1,1)).  Gradients of mean_i MSE_i (per-video backward accumulated, each scaled
    by 1/n: the packed-batch loss of the build's trainers, equal to the reference's own step when n == 1)."""
    opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-5)
    crit = torch.nn.MSELoss()
    opt.zero_grad()
    losses = []
    for x, tgt in videos:
        y = model(torch.from_numpy(x.copy()))
        loss = crit(y, torch.from_numpy(tgt)) / len(videos)
        loss.backward()
        losses.append(float(loss) * len(videos))
        out[f"{tag}/scores/T{x.shape[0]}"] = y.detach().numpy().reshape(-1)
    out[f"{tag}/losses"] = np.asarray(losses, dtype=np.float64)
    digest_into(out, f"{tag}/grad", [(k, p.grad) for k, p in model.named_parameters()])
    opt.step()
    digest_into(out, f"{tag}/param1", list(model.named_parameters()))


def video(T, D, seed):
    x = R.features(T, 1, D, seed) - 0.1            # mostly positive, a few negative entries
    tgt = np.random.default_rng(seed + 5).random((T, 1, 1)).astype(np.float32)
    return x, tgt


out = {}
D = 1024
t = lambda d: {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}

# ---- VASNet, one T = 300 video (the reference's own step) and a 3-video accumulated batch
for tag, lens in (("vasnet_T300", [300]), ("vasnet_batch3", [300, 163, 320])):
    m = VASNet(input_size=D).eval()
    m.load_state_dict(t(R.vasnet_weights(D, 31)))
    one_step(m, [video(T, D, 4000 + i) for i, T in enumerate(lens)], out, tag)
    print(tag, out[f"{tag}/losses"])

# ---- DSN (BiLSTM 1024 -> 2 x 256), T = 300
m = DSN(input_size=D, hidden_size=256).eval()
m.load_state_dict(t(R.lstm_weights("rnn.", D, 256, 1, 32, "out.0.")))
one_step(m, [video(300, D, 4100)], out, "dsn_T300")
print("dsn", out["dsn_T300/losses"])

# ---- sLSTM (2-layer BiLSTM 1024 -> 2 x 1024), T = 60
m = sLSTM(input_size=D, hidden_size=1024, num_layers=2).eval()
m.load_state_dict(t(R.lstm_weights("lstm.", D, 1024, 2, 33, "out.")))
one_step(m, [video(60, D, 4200)], out, "slstm_T60")
print("slstm", out["slstm_T60/losses"])

path = os.path.join(HERE, "train_full.npz")
np.savez_compressed(path, **out)
print(f"train_full: {os.path.getsize(path)/1024:.1f} KB, {len(out)} arrays")
