#!/usr/bin/env python3
"""Goldens of the reference's optional GRU cell: the REAL `summarizer.models.dsn.DSN(cell="gru")` (dsn.py:28-47, nn.GRU,
bidirectional) run here on CPU -- outputs for T in {1, 2, 37} and batch 1 / 3, and loss + every gradient of one MSE step.
Run once in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gru.py"""
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
from summarizer.models.dsn import DSN

out = {}
D, H = 64, 16
for L in (1, 2):
    torch.manual_seed(900 + L)
    m = DSN(input_size=D, hidden_size=H, num_layers=L, cell="gru")
    tag = f"L{L}"
    for k, v in m.state_dict().items():
        out[f"{tag}/w/{k}"] = v.numpy().copy()
    for T, B in ((1, 1), (2, 1), (37, 1), (37, 3)):
        x = R.features(T, B, D, 9000 + 10 * T + B) - 0.2
        with torch.no_grad():
            y = m(torch.from_numpy(x))
        out[f"{tag}/x/T{T}B{B}"] = x; out[f"{tag}/y/T{T}B{B}"] = y.numpy()
    # one MSE step's loss and gradients (T = 37, B = 1), input gradient included
    x = torch.from_numpy(R.features(37, 1, D, 9371) - 0.2).requires_grad_(True)
    tgt = torch.from_numpy(np.random.default_rng(5).random((37, 1, 1)).astype(np.float32))
    loss = torch.nn.functional.mse_loss(m(x), tgt)
    loss.backward()
    out[f"{tag}/loss"] = np.float32(loss.item()); out[f"{tag}/target"] = tgt.numpy(); out[f"{tag}/dx"] = x.grad.numpy().copy()
    for k, p in m.named_parameters():
        out[f"{tag}/grad/{k}"] = p.grad.numpy().copy()
path = os.path.join(HERE, "gru_small.npz")
np.savez_compressed(path, **out)
print(f"gru_small: {os.path.getsize(path)/1024:.1f} KB, {len(out)} arrays")
