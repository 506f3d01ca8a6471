#!/usr/bin/env python3
"""Goldens for SumGAN's forward-running LSTM stacks from the REAL reference modules (eLSTM, cLSTM; small sizes):
inputs, state_dict, outputs, and the gradients of a fixed scalar loss w.r.t. the input and every parameter.
-> tests/golden/sumgan_lstm.npz.  Run once in the build container:
   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sumgan.py"""
import os, sys, types
import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"; sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
for name in ["h5py", "ortools", "ortools.algorithms", "ortools.algorithms.pywrapknapsack_solver"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["ortools.algorithms"].pywrapknapsack_solver = sys.modules["ortools.algorithms.pywrapknapsack_solver"]
sys.path.insert(0, "/root/reference")
import torch
import summarizer.models.sumgan as ref

torch.set_num_threads(4)
D, H, L = 64, 32, 2
out = {"meta": np.array([D, H, L])}
rng = np.random.default_rng(17)

def record(tag, module, x, outputs_fn):
    x = torch.from_numpy(x).requires_grad_(True)
    outs = outputs_fn(module, x)
    loss = 0
    for i, o in enumerate(outs):
        cw = torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32))
        out[f"{tag}/cw{i}"] = cw.numpy(); out[f"{tag}/y{i}"] = o.detach().numpy()
        loss = loss + (o * cw).sum()
    loss.backward()
    out[f"{tag}/x"] = x.detach().numpy(); out[f"{tag}/dx"] = x.grad.numpy()
    for k, p in module.named_parameters():
        out[f"{tag}/w/{k}"] = p.detach().numpy().copy(); out[f"{tag}/g/{k}"] = p.grad.numpy().copy()

for T, B in ((37, 1), (20, 3)):
    torch.manual_seed(100 + T)
    e = ref.eLSTM(input_size=D, hidden_size=H, num_layers=L)
    record(f"elstm_T{T}B{B}", e, rng.standard_normal((T, B, D)).astype(np.float32) * 0.5,
           lambda m, x: (lambda r: [r[0][0], r[0][1], r[1]])(m(x)))
    c = ref.cLSTM(input_size=D, hidden_size=H, num_layers=L)
    record(f"clstm_T{T}B{B}", c, rng.standard_normal((T, B, D)).astype(np.float32) * 0.5, lambda m, x: list(m(x)))
    torch.manual_seed(200 + T)
    dl = ref.dLSTM(input_size=D, hidden_size=H, num_layers=L)
    h0 = torch.from_numpy(rng.standard_normal((L, B, H)).astype(np.float32) * 0.4).requires_grad_(True)
    c0 = torch.from_numpy(rng.standard_normal((L, B, H)).astype(np.float32) * 0.4).requires_grad_(True)
    tag = f"dlstm_T{T}B{B}"
    xh = dl(T, h0, c0)
    cw = torch.from_numpy(rng.standard_normal(tuple(xh.shape)).astype(np.float32))
    (xh * cw).sum().backward()
    out[f"{tag}/h0"] = h0.detach().numpy(); out[f"{tag}/c0"] = c0.detach().numpy(); out[f"{tag}/cw"] = cw.numpy()
    out[f"{tag}/y"] = xh.detach().numpy(); out[f"{tag}/dh0"] = h0.grad.numpy(); out[f"{tag}/dc0"] = c0.grad.numpy()
    for k, p in dl.named_parameters():
        out[f"{tag}/w/{k}"] = p.detach().numpy().copy(); out[f"{tag}/g/{k}"] = p.grad.numpy().copy()
np.savez_compressed(os.path.join(HERE, "sumgan_lstm.npz"), **out)
print(sorted(k for k in out if k.endswith("y0")), os.path.getsize(os.path.join(HERE, "sumgan_lstm.npz")) / 1024, "KB")
