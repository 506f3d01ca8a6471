import sys, os, ctypes as C
_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests", "golden"))
import numpy as np, torch, recipes as R
from summarizer_amd import _lib
from summarizer_amd.models.vasnet import VASNet
lib = _lib.load()
lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)]
m = VASNet().cuda().eval()
x = torch.from_numpy(np.concatenate([R.features(T, 1, 1024, i)[:, 0, :] for i, T in enumerate(lens)])).cuda()
with torch.no_grad():
    for _ in range(3): s = m.score_packed(x, lens)
torch.cuda.synchronize()
nb = 2048
out = np.zeros(nb * 4, dtype=np.uint64)
lib.sumk_prof_gemm_stamps(out.ctypes.data_as(C.POINTER(C.c_uint64)), nb)
o = out.reshape(nb, 4).astype(np.float64)[(760 if os.environ.get('SUMK_STAMP_TAG') is None else 0):]
o = o[o[:, 3] > 0]
tot, kl, ep, nt = o[:, 0], o[:, 1], o[:, 2], o[:, 3]
print(f"tag {os.environ.get('SUMK_STAMP_TAG', 'alpha.V (tail of the buffer)')}: blocks {len(o)} tiles/block {nt.min():.0f}-{nt.max():.0f} (mean {nt.mean():.2f}); block total median {np.median(tot):.0f} max {tot.max():.0f} min {tot.min():.0f}; k-loop {np.median(kl/tot)*100:.0f}% epilogue {np.median(ep/tot)*100:.0f}%; per tile: k-loop {np.median(kl/nt):.0f} epilogue {np.median(ep/nt):.0f}")
print("block total percentiles p1/p10/p50/p90/p99:", [int(np.percentile(tot, q)) for q in (1, 10, 50, 90, 99)])
print("mean T", np.mean(lens), "k-iters/tile ~", np.mean([ (t+31)//32 for t in lens]))
