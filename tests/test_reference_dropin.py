"""CONTAINER-ONLY (skipped where /root/reference is absent, i.e. on the GPU box): the reference's own, unedited
`summarizer/utils/config.py` (HParameters + the `-m <model>` registry, config.py:58-82) and `summarizer/main.py` resolve the
hot-path models to the HIP trainers once `summarizer_amd.install_as_reference()` has run, and those trainers construct
from a real `hps` up to `reset()` (no GPU call needed that far).  Runs in a subprocess so the aliases and the stubs for the
packages this image lacks (h5py, tensorboard's SummaryWriter; ortools is not needed at all any more) do not leak."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"

SCRIPT = r'''
import json, os, sys, types
import numpy as np
root, ref, tmp = sys.argv[1:4]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden")); sys.path.append(ref)
sys.dont_write_bytecode = True
# packages the reference imports and this image lacks (a real installation has them)
h5 = types.ModuleType("h5py"); sys.modules["h5py"] = h5
tb = types.ModuleType("torch.utils.tensorboard")
class SummaryWriter:
    def __init__(self, *a, **k): pass
    def add_scalar(self, *a, **k): pass
    def add_histogram(self, *a, **k): pass
    def add_hparams(self, *a, **k): pass
    def close(self): pass
tb.SummaryWriter = SummaryWriter
sys.modules["torch.utils.tensorboard"] = tb

import summarizer_amd
installed = summarizer_amd.install_as_reference()
assert "ortools" not in sys.modules
from summarizer.utils.config import HParameters          # the reference's file, unedited
import summarizer.main as ref_main                        # the reference's driver, unedited
import summarizer.models as ref_models                    # reference package (rand / logistic stay the reference's)
assert ref_main.train.__code__.co_varnames[0] == "hps"
assert os.path.realpath(sys.modules["summarizer.utils.config"].__file__).startswith(os.path.realpath(ref))

import recipes as R
from summarizer_amd.utils.datasets import DictDataset
os.chdir(tmp)
os.makedirs("splits", exist_ok=True)
videos = {f"video_{i+1}": R.synthetic_video(40 + 5 * i, 50 + i, n_users=3, D=64) for i in range(4)}
DictDataset(videos).save_npz(os.path.join(tmp, "summarizer_dataset_summe_google_pool5.npz"))
keys = list(videos)
json.dump([{"train_keys": keys[:3], "test_keys": keys[3:]}], open("splits/summe_splits.json", "w"))

out = {}
expected = {"vasnet": ("summarizer_amd.models.vasnet", "VASNetTrainer", {"local": "5", "input_size": "64"}),
            "dsn": ("summarizer_amd.models.dsn", "DSNTrainer", {"num_episodes": "3", "input_size": "64", "hidden_size": "16"}),
            "transformer": ("summarizer_amd.models.transformer", "TransformerTrainer", {"encoder_layers": "2", "input_size": "64"}),
            "sumgan": ("summarizer_amd.models.sumgan", "SumGANTrainer", {"input_size": "64", "hidden_size": "16"})}
for name, (module, cls, extra) in expected.items():
    hps = HParameters()
    hps.load_from_args({"model": name, "use_cuda": "no", "splits_files": ["splits/summe_splits.json"], "log_level": "error",
                        "datasets": os.path.join(tmp, "summarizer_dataset_summe_google_pool5.npz"), "extra_params": extra,
                        "epochs": 1})
    assert hps.model_class.__module__ == module and hps.model_class.__name__ == cls, hps.model_class
    trainer = hps.model_class(hps, "splits/summe_splits.json")          # main.py:25
    assert trainer.reset() is trainer                                    # main.py:27 chains .reset().train(fold)
    assert type(trainer.model).__module__.startswith("summarizer_amd.models")
    assert trainer._get_train_test_keys(0) == (keys[:3], keys[3:])
    assert os.path.exists(os.path.join(hps.log_path, os.path.basename(sys.modules[module].__file__)))   # config.py:165-167
    for m in ("train", "test", "predict_dataset", "save_best_weights", "load_weights"):
        assert callable(getattr(trainer, m))
    try:
        trainer.save_best_weights(os.path.join(tmp, "w.pth"))
        raise SystemExit("save_best_weights must raise before training")
    except Exception as e:
        assert "best_weights" in str(e)
    out[name] = [hps.model_class.__module__, type(trainer.model).__name__, sum(p.numel() for p in trainer.model.parameters())]
# out-of-scope models still come from the reference checkout
hps = HParameters(); hps.load_from_args({"model": "logistic", "use_cuda": "no", "splits_files": ["splits/summe_splits.json"], "log_level": "error",
                                         "datasets": os.path.join(tmp, "summarizer_dataset_summe_google_pool5.npz"), "extra_params": {}})
assert hps.model_class.__module__ == "summarizer.models.logistic"
out["installed"] = installed
print("RESULT " + json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "summarizer")), reason="reference checkout not present (GPU box)")
def test_reference_config_and_main_resolve_to_hip_trainers(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, REF, str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["vasnet"][:2] == ["summarizer_amd.models.vasnet", "VASNet"]
    assert out["dsn"][:2] == ["summarizer_amd.models.dsn", "DSN"]
    assert out["transformer"][1] == "Transformer" and out["sumgan"][1] == "SumGAN"
    assert len(out["installed"]) == 6


def test_install_as_reference_refuses_late_install():
    code = ("import sys, types; sys.modules['summarizer.models.vasnet'] = types.ModuleType('summarizer.models.vasnet');"
            f"sys.path.insert(0, {ROOT!r}); import summarizer_amd\n"
            "try:\n    summarizer_amd.install_as_reference()\nexcept ImportError as e:\n    print('OK', 'before importing' in str(e))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.stdout.strip() == "OK True", r.stdout + r.stderr
