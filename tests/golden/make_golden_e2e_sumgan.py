#!/usr/bin/env python3
"""G9: end-to-end golden from the REAL reference SumGANTrainer (VAE pre-training, then selector+encoder / decoder /
discriminator updates with three Adam optimisers; supervised sparsity, input noise in epoch 0), small model, synthetic
SumMe-shaped dataset, in-memory h5py stand-in.  torch.randn_like / torch.rand (reparameterisation, uniform scores, noise)
are replaced by recipes.DetRandom so the HIP trainer can be fed the same draws.
-> tests/golden/e2e_sumgan.npz.   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_e2e_sumgan.py"""
import os, sys, types, random
import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"; sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from summarizer_amd.utils.datasets import synthetic_dataset
from summarizer_amd.utils.hps import make_hps
import recipes as R

DS = {}
h5 = types.ModuleType("h5py")
h5.File = lambda path, mode="r": DS[path]
sys.modules["h5py"] = h5
for name in ["ortools", "ortools.algorithms", "ortools.algorithms.pywrapknapsack_solver"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["ortools.algorithms"].pywrapknapsack_solver = sys.modules["ortools.algorithms.pywrapknapsack_solver"]
sys.path.insert(0, "/root/reference")
import torch
import summarizer.models.sumgan as ref

torch.set_num_threads(4)
D, SEED = 64, 123
EP = {"input_size": str(D), "sLSTM_hidden_size": "32", "edLSTM_hidden_size": "48", "cLSTM_hidden_size": "32",
      "pretrain_vae": "1", "epoch_noise": "1", "sup": True}
ds = synthetic_dataset(9, seed=8, D=D, t_range=(30, 60), n_users=5)
keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))
DS["synthetic.h5"] = ds
hps = make_hps("synthetic.h5", [{"train_keys": keys[3:], "test_keys": keys[:3]}], epochs=2, test_every_epochs=1, lr=1e-3,
               use_cuda=False, selection_algorithm="rank", extra_params=dict(EP))
torch.manual_seed(SEED); random.seed(SEED)
tr = ref.SumGANTrainer(hps, hps.splits_files[0]).reset()
w0 = {k: v.detach().numpy().copy() for k, v in tr.model.state_dict().items()}
with R.DetRandom(SEED).patch() as det:
    best = tr.train(0)
    n_draws = det.n
out = {f"w0/{k}": v for k, v in w0.items()}
out.update({f"w1/{k}": v.detach().numpy().copy() for k, v in tr.model.state_dict().items()})
sc = hps.writer.scalars
for t in ("Lse", "Ld", "Lc", "D_x", "D_x_hat", "D_x_hat_p"):
    out[t] = np.array([v for _, v in sc[f"synthetic/Fold_1/Train/{t}"]], dtype=np.float64)
out["corr"] = np.array([v for _, v in sc["synthetic/Fold_1/Test/Correlation"]], dtype=np.float64)
out["f_avg"] = np.array([v for _, v in sc["synthetic/Fold_1/Test/F-score_avg"]], dtype=np.float64)
out["f_max"] = np.array([v for _, v in sc["synthetic/Fold_1/Test/F-score_max"]], dtype=np.float64)
out["best"] = np.array(best, dtype=np.float64)
out["meta"] = np.array([D, SEED, 9, 8, 30, 60, 5, n_draws])
tr.model.eval()
with torch.no_grad():
    for k in keys[:3]:
        out[f"scores/{k}"] = tr.model(torch.from_numpy(ds[k]["features"][...]).unsqueeze(1)).squeeze().numpy()
np.savez_compressed(os.path.join(HERE, "e2e_sumgan.npz"), **out)
print({t: out[t] for t in ("Lse", "Ld", "Lc", "D_x", "D_x_hat", "D_x_hat_p", "corr", "f_avg")}, "draws", n_draws)
print(os.path.getsize(os.path.join(HERE, "e2e_sumgan.npz")) / 1024, "KB")
