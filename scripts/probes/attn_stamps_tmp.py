import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipes as R
from summarizer_amd import kernels, _lib
from summarizer_amd.autograd import VasnetFunction
from summarizer_amd.models.vasnet import VASNet
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
D = 1024; lens = [int(t) for t in rng.integers(150, 321, size=50)]
w = R.vasnet_weights(D, 77)
m = VASNet(input_size=D); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
x = torch.from_numpy(np.concatenate([R.features(T, 1, D, 300 + i)[:, 0, :] for i, T in enumerate(lens)]) - 0.1).to(dev)
sb = kernels.SeqBatch.get(lens, dev)
opts = dict(scale=float(m.scale), eps=1e-6, ignore_self=False, aperture=None, dropout_p=0.5, seed=99, precision="bf16")
names = [k for _, k in kernels.VASNET_FIELDS]; params = dict(m.named_parameters())
for it in range(5):
    for p in params.values(): p.grad = None
    s = VasnetFunction.apply(x, sb, opts, None, None, names, *[params[n] for n in names])
    s.sum().backward()
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((2, 512, 8), dtype=np.uint64)
rc = lib.sumk_attn_stamps_tmp(buf.ctypes.data_as(C.c_void_p)); assert rc == 0
for d, name in ((0, "fwd"), (1, "bwd")):
    b = buf[d].astype(np.int64); ok = b[:, 6] == 5
    b = b[ok]; print(name, "blocks NJ=5:", ok.sum())
    t = (b[:, 1:6] - b[:, 0:5]) / 100.0   # s_memtime 100 MHz -> us
    print("  us: gemm1 %.1f  rowop %.1f  barrier %.1f  gemm2-pass0 %.1f  gemm2-rest %.1f  total %.1f" % (*t.mean(0), (b[:, 5] - b[:, 0]).mean() / 100.0))
    print("  start spread us %.1f  end spread %.1f" % ((b[:, 0].max() - b[:, 0].min()) / 100.0, (b[:, 5].max() - b[:, 0].min()) / 100.0))
