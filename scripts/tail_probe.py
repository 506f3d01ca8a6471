"""Probe: where does Trainer.test() spend its wall time once scoring runs on the HIP path?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from summarizer_amd.utils.datasets import synthetic_dataset
from summarizer_amd.utils.hps import make_hps
from summarizer_amd.models.vasnet import VASNetTrainer
from summarizer_amd.utils import eval as E
ds = synthetic_dataset(50, seed=1, D=1024, t_range=(150, 320), n_users=20)
keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))
hps = make_hps(ds, [{"train_keys": keys[:1], "test_keys": keys}])
tr = VASNetTrainer(hps, hps.splits_files[0]).reset()
tr.model.eval()
for _ in range(2): tr.test(0)
t0 = time.perf_counter(); tr.test(0); t_all = time.perf_counter() - t0
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter(); s = tr._score_keys(keys); torch.cuda.synchronize(); t_score = time.perf_counter() - t0
t0 = time.perf_counter(); c_np = tr._eval_scores(s, keys); t_corr = time.perf_counter() - t0
t0 = time.perf_counter(); f_np = tr._eval_summary(s, keys); t_sum = time.perf_counter() - t0
t0 = time.perf_counter(); corr, fa, fm, _ = tr._evaluate_native(s, keys); t_nat = time.perf_counter() - t0
print(f"Trainer.test on 50 videos: total {t_all*1e3:.1f} ms | scoring incl. D2H {t_score*1e3:.1f} ms | native eval tail {t_nat*1e3:.1f} ms "
      f"(numpy path: rank-correlation {t_corr*1e3:.1f} ms + summary/F-score {t_sum*1e3:.1f} ms) | native == numpy: "
      f"{abs(np.mean(corr) - c_np) < 1e-12 and abs(np.mean(fa) - f_np[0]) < 1e-6 and abs(np.mean(fm) - f_np[1]) < 1e-6}")
