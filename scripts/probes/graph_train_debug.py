"""debug: VASNetTrainer losses per epoch with / without per-video HIP graphs"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from summarizer_amd.models.vasnet import VASNetTrainer
from summarizer_amd.utils.datasets import synthetic_dataset
from summarizer_amd.utils.hps import make_hps
flag, local = sys.argv[1], sys.argv[2]
ds = synthetic_dataset(11, seed=5, D=128, t_range=(40, 90), n_users=6)
keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))
ep = {"input_size": "128", "hip_graph": flag}
if local != "0":
    ep["local"] = local
hps = make_hps(ds, [{"train_keys": keys[3:], "test_keys": keys[:3]}], epochs=5, test_every_epochs=int(sys.argv[3]) if len(sys.argv) > 3 else 1, lr=1e-3, extra_params=ep)
torch.manual_seed(7); random.seed(3)
tr = VASNetTrainer(hps, hps.splits_files[0]).reset()
tr.model.dropout.p = 0.0
random.seed(3)
_orig = tr.test
def _test(fold):
    torch.cuda.synchronize()
    print("  params checksum", float(tr.optimizer.flat_param.double().abs().sum()), "adam steps", tr.optimizer._state.tolist()[0])
    return _orig(fold)
tr.test = _test
tr.train(0)
print(flag, local, [round(v, 6) for _, v in hps.writer.scalars["synthetic/Fold_1/Train/Loss"]])
