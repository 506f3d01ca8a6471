#!/usr/bin/env python3
"""Fused attention strips of the bf16-source training step (csrc/attn_b16.hip) against the separate launches they replace (GEMM -> softmax
kernel -> GEMM): training-mode scores with dropout and every parameter gradient, with and without dX, at D = 1024 for (a) fifty TVSum-sized
videos, (b) a ragged batch with T = 1 ... 320, (c) local attention + ignore_self, (d) no dropout, and at D = 256 / 512 / 2048.
Written to an .npz; run once with SUMK_ATTN_FUSED=1 (default) and once with SUMK_ATTN_FUSED=0 and compare (tests/test_gpu_train_full.py).
usage: python scripts/probes/attn_fused_equiv.py out.npz"""
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
rng = np.random.default_rng(5)
cases = [("tvsum", 1024, [int(t) for t in rng.integers(150, 321, size=50)], {}, 0.5),
         ("ragged", 1024, [1, 37, 64, 65, 200, 320, 128, 31, 319, 257, 2, 63] + [int(t) for t in rng.integers(60, 321, size=44)], {}, 0.5),
         ("local", 1024, [int(t) for t in rng.integers(100, 321, size=46)], dict(ignore_self=True, attention_aperture=40), 0.5),
         ("nodrop", 1024, [300, 129, 64] + [int(t) for t in rng.integers(200, 321, size=32)], {}, 0.0),
         # other widths: k-tiles per strip 4 / 8 / 32 (fewer resp. more than the 16 accumulator registers the keep bits are spread over),
         # 1 / 2 / 8 column passes of the second product
         ("d256", 256, [320, 1, 65] + [int(t) for t in rng.integers(120, 321, size=150)], {}, 0.5),
         ("d512", 512, [int(t) for t in rng.integers(100, 321, size=84)], dict(ignore_self=True), 0.5),
         ("d2048", 2048, [int(t) for t in rng.integers(100, 321, size=24)], {}, 0.5)]
for tag, D, lens, kw, p_drop in cases:
    w = R.vasnet_weights(D, 77)
    m = VASNet(input_size=D, **kw); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
    x = torch.from_numpy(np.concatenate([R.features(T, 1, D, 300 + i)[:, 0, :] for i, T in enumerate(lens)]) - 0.1).to(dev)
    sb = kernels.SeqBatch.get(lens, dev)
    assert sum(lens) * D >= 8192 * 1024, (tag, sum(lens))       # (the bf16-source step needs >= 512 row tiles of 128 x 128)
    opts = dict(scale=float(m.scale), eps=1e-6, ignore_self=bool(m.ignore_self), aperture=m.aperture, dropout_p=p_drop, seed=99, precision="bf16")
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
