import sys, os, time
_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests", "golden"))
import numpy as np, torch, recipes as R
from summarizer_amd.models.vasnet import VASNet
torch.manual_seed(0)
lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)]
m = VASNet().cuda().eval()
x = torch.from_numpy(np.concatenate([R.features(T, 1, 1024, i)[:, 0, :] for i, T in enumerate(lens)])).cuda()
def run(prec, fold):
    m.precision = prec; m.fold_vo = fold
    with torch.no_grad():
        for _ in range(10): s = m.score_packed(x, lens)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): s = m.score_packed(x, lens)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 10, s
ms0, s0 = run("fp32", False)
for prec, fold in (("fp32", True), ("bf16x6", False), ("bf16x6", True), ("bf16x3", True)):
    ms, s = run(prec, fold)
    print(f"{prec} fold={fold}: {ms:.4f} ms  max|d| vs fp32 default {float((s - s0).abs().max()):.2e}")
print(f"fp32 default {ms0:.4f} ms")
