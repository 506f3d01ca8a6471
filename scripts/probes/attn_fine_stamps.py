#!/usr/bin/env python3
"""In-kernel stamps of ONE per-video GEMM launch of the headline batch (diagnostic build only):
   SUMK_LIB_PATH=$PWD/summarizer_amd/libsumk_diag.so SUMK_GEMM_DBG=6 SUMK_STAMP_TAG=3 python scripts/probes/attn_fine_stamps.py
(tag 3 = Q.K^T, 4 = alpha.V).  Prints the block-duration percentiles, the k-loop shares of wave 0 (barrier 1 / wait + LDS write /
barrier 2 / load issue + fragment reads + MFMAs), the launch's wall-clock window and how many blocks shared a CU."""
import sys, os, ctypes as C, collections
_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests", "golden"))
import numpy as np, torch, recipes as R
from summarizer_amd import _lib
from summarizer_amd.models.vasnet import VASNet
lib = _lib.load()
lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)]
m = VASNet().cuda().eval()
x = torch.from_numpy(np.concatenate([R.features(T, 1, 1024, i)[:, 0, :] for i, T in enumerate(lens)])).cuda()
with torch.no_grad():
    for _ in range(5): s = m.score_packed(x, lens)
torch.cuda.synchronize()
nb = 2048
out = np.zeros(nb * 4, dtype=np.uint64)
_lib.check(lib.sumk_prof_gemm_stamps(out.ctypes.data_as(C.POINTER(C.c_uint64)), nb), "stamps")
fine = np.zeros(nb * 8, dtype=np.uint64)
_lib.check(lib.sumk_prof_gemm_stamps(fine.ctypes.data_as(C.POINTER(C.c_uint64)), -nb), "fine stamps")
o = out.reshape(nb, 4).astype(np.float64); f = fine.reshape(nb, 8)
live = o[:, 3] > 0
o, f = o[live], f[live]
tot, kl, ep, nt = o[:, 0], o[:, 1], o[:, 2], o[:, 3]
print(f"tag {os.environ.get('SUMK_STAMP_TAG')}: blocks {len(o)}, tiles/block {nt.min():.0f}-{nt.max():.0f}; block total p1/p10/p50/p90/p99 =",
      [int(np.percentile(tot, q)) for q in (1, 10, 50, 90, 99)], f"k-loop {np.median(kl/tot)*100:.0f}% epilogue {np.median(ep/tot)*100:.0f}%")
sh = f[:, :4].astype(np.float64)
tot_sh = sh.sum(axis=1)
print("k-loop shares of wave 0 (median over blocks): barrier1 %.1f%%  wait+LDS write %.1f%%  barrier2 %.1f%%  issue+reads+MFMA %.1f%%; cycles per k-iteration (median) %.0f" %
      tuple(list(np.median(sh / tot_sh[:, None], axis=0) * 100) + [np.median(tot_sh / (nt * 32))]))
t0, t1 = f[:, 4].astype(np.int64), f[:, 5].astype(np.int64)
print(f"wall-clock: first start -> last end {(t1.max() - t0.min()) / 100.0:.1f} us; block durations p10/p50/p90 = "
      f"{np.percentile(t1 - t0, 10) / 100:.1f}/{np.percentile(t1 - t0, 50) / 100:.1f}/{np.percentile(t1 - t0, 90) / 100:.1f} us; "
      f"start spread {(t0.max() - t0.min()) / 100.0:.1f} us; shader clock ~ {np.median(tot / ((t1 - t0) / 100.0)) / 1e3:.2f} GHz")
hw, xcc = f[:, 6].astype(np.int64), f[:, 7].astype(np.int64) & 0xF
cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
per = collections.Counter(cu.tolist())
hist = collections.Counter(per.values())
print("blocks per CU:", dict(sorted(hist.items())), "CUs used:", len(per))
dur = (t1 - t0) / 100.0
for k in sorted(hist):
    sel = np.array([per[c] == k for c in cu.tolist()])
    print(f"  CUs with {k} blocks: block duration median {np.median(dur[sel]):.1f} us, max {dur[sel].max():.1f} us; cycles median {np.median(tot[sel]):.0f}")
