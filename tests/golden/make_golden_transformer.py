#!/usr/bin/env python3
"""Golden vectors for the Transformer-encoder scorer from the REAL reference (summarizer/models/transformer.py), eval mode.
Run once in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_transformer.py"""
import os, sys, types, json
import numpy as np
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"; sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE)
for name in ["h5py", "ortools", "ortools.algorithms", "ortools.algorithms.pywrapknapsack_solver"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["ortools.algorithms"].pywrapknapsack_solver = sys.modules["ortools.algorithms.pywrapknapsack_solver"]
sys.path.insert(0, "/root/reference")
import torch
import recipes as R
from summarizer.models.transformer import Transformer
torch.set_num_threads(4)

out, meta = {}, {}
cases = {"small": dict(input_size=64, encoder_layers=2, attention_heads=4),
         "small_res": dict(input_size=64, encoder_layers=3, attention_heads=8, more_residuals=True, epsilon=1e-3),
         "small_pos": dict(input_size=64, encoder_layers=1, attention_heads=2, max_length=64, pos_embed="simple")}
for ci, (name, kw) in enumerate(cases.items()):
    m = Transformer(**kw).eval()
    w = R.transformer_weights(kw["input_size"], kw["encoder_layers"], 600 + ci, max_length=kw.get("max_length"))
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not missing.unexpected_keys, missing
    assert all(k.startswith("transformer_encoder_layer.") for k in missing.missing_keys), missing.missing_keys   # the unused prototype layer
    for k, v in w.items(): out[f"{name}/w/{k}"] = v
    for T in ([1, 2, 37] if kw.get("max_length") else [1, 2, 37, 130]):
        for B in ([1, 3] if T == 37 else [1]):
            x = R.features(T, B, kw["input_size"], 6000 + 10 * ci + T + B) - 0.2
            out[f"{name}/x/T{T}B{B}"] = x
            with torch.no_grad():
                out[f"{name}/y/T{T}B{B}"] = m(torch.from_numpy(x.copy())).numpy()
    meta[name] = kw
out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
np.savez_compressed(os.path.join(HERE, "transformer_small.npz"), **out)
print("small", os.path.getsize(os.path.join(HERE, "transformer_small.npz")) / 1024, "KB")

out = {}
for ci, (T, B, layers) in enumerate([(200, 1, 6), (61, 2, 2)]):
    D = 1024
    m = Transformer(input_size=D, encoder_layers=layers, attention_heads=8).eval()
    w = R.transformer_weights(D, layers, 9000 + ci)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    x = R.features(T, B, D, 9100 + ci)
    with torch.no_grad():
        out[f"c{ci}/y"] = m(torch.from_numpy(x.copy())).numpy()
    out[f"c{ci}/cfg"] = np.frombuffer(json.dumps(dict(T=T, B=B, D=D, layers=layers, heads=8, wseed=9000 + ci, xseed=9100 + ci,
                                                     wdigest=R.digest(w))).encode(), dtype=np.uint8)
np.savez_compressed(os.path.join(HERE, "transformer_full.npz"), **out)
print("full", os.path.getsize(os.path.join(HERE, "transformer_full.npz")) / 1024, "KB")

# ----------------------------------------------------------------------------- training steps (all dropouts forced to 0)
def no_dropout(m):
    m.dropout.p = 0.0
    for lyr in m.transformer_encoder.layers:
        lyr.dropout.p = lyr.dropout1.p = lyr.dropout2.p = 0.0
        lyr.self_attn.dropout = 0.0
    return m

out = {}
T, D = 37, 64
tgt = np.random.default_rng(5).random((T, 1, 1)).astype(np.float32)
x = R.features(T, 1, D, 4242) - 0.2
for name, kw in {"tf": dict(input_size=D, encoder_layers=2, attention_heads=4),
                 "tf_res": dict(input_size=D, encoder_layers=1, attention_heads=8, more_residuals=True)}.items():
    m = no_dropout(Transformer(**kw)).train()
    w = R.transformer_weights(D, kw["encoder_layers"], 4100)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    used = [(k, p) for k, p in m.named_parameters() if not k.startswith("transformer_encoder_layer.") and not k.startswith("transformer_encoder.norm.")]
    opt = torch.optim.Adam([p for _, p in used], lr=5e-5, weight_decay=1e-5)
    for s in range(3):
        loss = torch.nn.functional.mse_loss(m(torch.from_numpy(x.copy())), torch.from_numpy(tgt))
        opt.zero_grad(); loss.backward()
        if s == 0:
            out[f"{name}/loss0"] = np.float32(loss.item())
            for k, p in used: out[f"{name}/grad0/{k}"] = p.grad.numpy().copy()
        opt.step()
        if s in (0, 2):
            for k, p in used: out[f"{name}/param{s+1}/{k}"] = p.detach().numpy().copy()
    for k, v in w.items(): out[f"{name}/w/{k}"] = v
out["x"] = x; out["target"] = tgt
np.savez_compressed(os.path.join(HERE, "transformer_train.npz"), **out)
print("train", os.path.getsize(os.path.join(HERE, "transformer_train.npz")) / 1024, "KB")
