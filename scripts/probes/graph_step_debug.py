"""debug: one captured VASNet training step replayed vs the same step run eagerly, from the same start"""
import os, sys, random, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from summarizer_amd.models.vasnet import VASNetTrainer
from summarizer_amd.training import FlatAdam
from summarizer_amd.utils.datasets import synthetic_dataset
from summarizer_amd.utils.hps import make_hps
ds = synthetic_dataset(11, seed=5, D=128, t_range=(40, 90), n_users=6)
keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))
hps = make_hps(ds, [{"train_keys": keys[3:], "test_keys": keys[:3]}], epochs=1, lr=1e-3, extra_params={"input_size": "128"})
def fresh():
    torch.manual_seed(7)
    tr = VASNetTrainer(hps, hps.splits_files[0]).reset()
    tr.model.dropout.p = 0.0
    tr.model.train()
    tr.optimizer = FlatAdam(tr.model.parameters(), lr=1e-3, weight_decay=1e-5)
    tr.model.graph_seed = None
    return tr
dev = torch.device("cuda:0")
# eager: three steps on keys 5, 6, 5
a = fresh()
la = [float(a._single_video_step(k, dev)[0]) for k in (keys[5], keys[6], keys[5], keys[6], keys[6], keys[5], keys[5], keys[6])]
pa = a.optimizer.flat_param.clone(); ga = a.optimizer.flat_grad.clone()
# graphs: eager step 5, 6 (epoch 0), then capture + replay 5, capture + replay 6
b = fresh()
lb = [float(b._single_video_step(k, dev)[0]) for k in (keys[5], keys[6])]
pool = None
ents = {}
for k in (keys[5], keys[6]):
    ents[k] = b._capture_step(k, dev, pool); pool = ents[k][0].pool()
    print("after capture", k, "param moved by", float((b.optimizer.flat_param - a.optimizer.flat_param).abs().max()), "state", b.optimizer._state.tolist())
    ents[k][0].replay(); torch.cuda.synchronize()
    lb.append(float(ents[k][1]))
for k in (keys[6], keys[5], keys[5], keys[6]):
    ents[k][0].replay(); torch.cuda.synchronize()
    lb.append(float(ents[k][1]))
pb = b.optimizer.flat_param.clone(); gb = b.optimizer.flat_grad.clone()
print("losses eager", la); print("losses graph", lb)
print("param diff", float((pa - pb).abs().max()), "grad diff", float((ga - gb).abs().max()), "grad max", float(ga.abs().max()))
print("adam state", a.optimizer._state.tolist(), b.optimizer._state.tolist())
