#!/usr/bin/env python3
"""G8: end-to-end golden from the REAL reference DSNTrainer (REINFORCE + supervised BCE + length penalty).
Runs summarizer.models.dsn.DSNTrainer (reset -> train -> test) on a synthetic SumMe-shaped dataset through an in-memory
h5py stand-in.  The Bernoulli samples are torch-CPU-RNG draws no other implementation can regenerate, so every sampled
action vector is RECORDED (in call order) and committed with the vector; the HIP trainer replays them.
Commits initial / final weights, per-epoch losses and rewards, per-step losses, the sampled actions, test scores and the
returned metrics -> tests/golden/e2e_dsn.npz.
Run once in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_e2e_dsn.py"""
import os, sys, types, random
import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"; sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from summarizer_amd.utils.datasets import synthetic_dataset
from summarizer_amd.utils.hps import make_hps

DS = {}
h5 = types.ModuleType("h5py")
h5.File = lambda path, mode="r": DS[path]
sys.modules["h5py"] = h5
for name in ["ortools", "ortools.algorithms", "ortools.algorithms.pywrapknapsack_solver"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["ortools.algorithms"].pywrapknapsack_solver = sys.modules["ortools.algorithms.pywrapknapsack_solver"]
sys.path.insert(0, "/root/reference")
import torch
import summarizer.models.dsn as ref_dsn

torch.set_num_threads(4)
D, H, SEED = 128, 32, 91
ds = synthetic_dataset(10, seed=6, D=D, t_range=(40, 80), n_users=5)
keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))
DS["synthetic.h5"] = ds
splits = [{"train_keys": keys[3:], "test_keys": keys[:3]}]
hps = make_hps("synthetic.h5", splits, epochs=3, test_every_epochs=1, lr=1e-3, use_cuda=False, selection_algorithm="rank",
               extra_params={"beta": "1", "sup": True, "num_episodes": "3"})

ACTIONS, STEP_LOSS = [], []

class RecordingBernoulli(ref_dsn.Bernoulli):
    def sample(self, *a, **k):
        s = super().sample(*a, **k)
        ACTIONS.append(s.reshape(-1).numpy().astype(np.uint8))
        return s
ref_dsn.Bernoulli = RecordingBernoulli

class RefTrainer(ref_dsn.DSNTrainer):
    def _init_model(self):
        super()._init_model()                      # parses beta / eps / num_episodes / sup exactly as the reference does
        torch.manual_seed(SEED)                    # (the default-size DSN() built above consumed generator state)
        return ref_dsn.DSN(input_size=D, hidden_size=H)

torch.manual_seed(SEED); random.seed(SEED)
tr = RefTrainer(hps, hps.splits_files[0]).reset()
w0 = {k: v.detach().numpy().copy() for k, v in tr.model.state_dict().items()}
best = tr.train(0)
out = {f"w0/{k}": v for k, v in w0.items()}
out.update({f"w1/{k}": v.detach().numpy().copy() for k, v in tr.model.state_dict().items()})
sc = hps.writer.scalars
out["losses"] = np.array([v for _, v in sc["synthetic/Fold_1/Train/Loss"]], dtype=np.float64)
out["rewards"] = np.array([v for _, v in sc["synthetic/Fold_1/Train/Reward"]], dtype=np.float64)
out["corr"] = np.array([v for _, v in sc["synthetic/Fold_1/Test/Correlation"]], dtype=np.float64)
out["f_avg"] = np.array([v for _, v in sc["synthetic/Fold_1/Test/F-score_avg"]], dtype=np.float64)
out["f_max"] = np.array([v for _, v in sc["synthetic/Fold_1/Test/F-score_max"]], dtype=np.float64)
out["best"] = np.array(best, dtype=np.float64)
out["n_actions"] = np.array([len(ACTIONS)])
out["actions_len"] = np.array([len(a) for a in ACTIONS], dtype=np.int32)
out["actions"] = np.concatenate(ACTIONS)
tr.model.eval()
with torch.no_grad():
    for k in keys[:3]:
        out[f"scores/{k}"] = tr.model(torch.from_numpy(ds[k]["features"][...]).unsqueeze(1)).squeeze().numpy()
out["meta"] = np.array([D, H, SEED, 10, 6, 40, 80, 5, 3])
np.savez_compressed(os.path.join(HERE, "e2e_dsn.npz"), **out)
print("beta", tr.beta, "sup", tr.sup, "episodes", tr.num_episodes)
print("losses", out["losses"], "rewards", out["rewards"], "corr", out["corr"], "f", out["f_avg"], out["f_max"], "best", best)
print(len(ACTIONS), "action vectors;", os.path.getsize(os.path.join(HERE, "e2e_dsn.npz")) / 1024, "KB")
