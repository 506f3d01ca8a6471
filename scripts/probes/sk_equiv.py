#!/usr/bin/env python3
"""Small-batch (in-launch split-K) path of VASNet against the large-batch kernels: scores in eval and training mode (dropout on),
every parameter gradient, with and without dX, for (a) ONE TVSum-sized video at D = 1024 -- the reference's calling pattern --
(b) a ragged batch with T = 1 ... 333 at D = 256, (c) three videos at D = 1024 with local attention + ignore_self, (d) D = 400 (K tails in every slice).  Written to an
.npz; run once with SUMK_SK=1 (default) and once with SUMK_SK=0 and compare (tests/test_gpu_vasnet.py).
usage: python scripts/probes/sk_equiv.py out.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipes as R
from summarizer_amd import kernels
from summarizer_amd.autograd import VasnetFunction
from summarizer_amd.models.vasnet import VASNet

dev = torch.device("cuda:0")
out = {}
cases = [("one", 1024, [300], {}), ("ragged", 256, [1, 37, 64, 65, 200, 333, 128, 31], {}),
         ("three", 1024, [211, 320, 150], dict(ignore_self=True, attention_aperture=40)),
         ("odd", 400, [100, 45, 129], {})]       # D = 400: K tails in every slice set, 4 + 2 slabs
for tag, D, lens, kw in cases:
    w = R.vasnet_weights(D, 77)
    m = VASNet(input_size=D, **kw); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
    x = torch.from_numpy(np.concatenate([R.features(T, 1, D, 300 + i)[:, 0, :] for i, T in enumerate(lens)]) - 0.1).to(dev)
    m.eval()
    with torch.no_grad():
        out[f"{tag}_scores_eval"] = m.score_packed(x, lens).cpu().numpy()
        out[f"{tag}_scores_eval_again"] = m.score_packed(x, lens).cpu().numpy()
    sb = kernels.SeqBatch.get(lens, dev)
    opts = dict(scale=float(m.scale), eps=1e-6, ignore_self=bool(m.ignore_self), aperture=m.aperture, dropout_p=0.5, seed=99, precision="fp32")
    names = [k for _, k in kernels.VASNET_FIELDS]
    params = dict(m.named_parameters())
    for want_dx in (False, True):
        for p in params.values():
            p.grad = None
        xg = x.clone().requires_grad_(want_dx)
        s = VasnetFunction.apply(xg, sb, opts, None, None, names, *[params[n] for n in names])
        (s * torch.linspace(-1, 1, s.numel(), device=dev)).sum().backward()
        sfx = "_dx" if want_dx else ""
        out[f"{tag}_scores_train{sfx}"] = s.detach().cpu().numpy()
        if want_dx:
            out[f"{tag}_dx"] = xg.grad.cpu().numpy()
        for n in names:
            out[f"{tag}_g{sfx}_" + n] = params[n].grad.cpu().numpy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
