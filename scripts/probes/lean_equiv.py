#!/usr/bin/env python3
"""Scores and every gradient of a ragged VASNet batch (T in 1 .. 333, D = 256; training mode with dropout), plus plain NT / NN GEMMs
with awkward shapes, written to an .npz -- run once with SUMK_LEAN=1 and once with SUMK_LEAN=0: the lean 64x64 kernel
(csrc/gemm_lean.hip) must reproduce the generic register-staged kernel bit for bit (same k order, one fmaf chain per element).
usage: python scripts/probes/lean_equiv.py out.npz"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipes as R
from summarizer_amd import _lib, kernels
from summarizer_amd.autograd import VasnetFunction
from summarizer_amd.models.vasnet import VASNet

lib = _lib.load()
dev = torch.device("cuda:0")
out = {}
D, lens = 256, [1, 37, 64, 65, 200, 333, 128, 31]
w = R.vasnet_weights(D, 77)
m = VASNet(input_size=D); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
x = torch.from_numpy(np.concatenate([R.features(T, 1, D, 300 + i)[:, 0, :] for i, T in enumerate(lens)]) - 0.1).to(dev)
m.eval()
with torch.no_grad():
    out["scores_eval"] = m.score_packed(x, lens).cpu().numpy()
xg = x.clone().requires_grad_(True)
sb = kernels.SeqBatch.get(lens, dev)
opts = dict(scale=float(m.scale), eps=1e-6, ignore_self=False, aperture=None, dropout_p=0.5, seed=99, precision="fp32")
names = [k for _, k in kernels.VASNET_FIELDS]
params = dict(m.named_parameters())
s = VasnetFunction.apply(xg, sb, opts, None, None, names, *[params[n] for n in names])
(s * torch.linspace(-1, 1, s.numel(), device=dev)).sum().backward()
out["scores_train"] = s.detach().cpu().numpy(); out["dx"] = xg.grad.cpu().numpy()
for n in names:
    out["g_" + n] = params[n].grad.cpu().numpy()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cpu").manual_seed(5)
for (M, N, K) in [(4, 4, 4), (64, 64, 32), (50, 64, 33), (64, 37, 100), (33, 60, 1024), (64, 64, 31)]:
    a = torch.randn(M, K, generator=g).to(dev); bt = torch.randn(N, K, generator=g).to(dev); bn = torch.randn(K, (N + 3) // 4 * 4, generator=g).to(dev)
    c = torch.empty(M, N, device=dev)
    if K % 4 == 0:
        _lib.check(lib.sumk_gemm_nt(a.data_ptr(), bt.data_ptr(), c.data_ptr(), M, N, K, st), "nt"); out[f"nt_{M}_{N}_{K}"] = c.cpu().numpy()
    Nn = bn.shape[1]
    c2 = torch.empty(M, Nn, device=dev)
    if K % 4 == 0:
        _lib.check(lib.sumk_gemm_nn(a.data_ptr(), bn.data_ptr(), c2.data_ptr(), M, Nn, K, st), "nn"); out[f"nn_{M}_{Nn}_{K}"] = c2.cpu().numpy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
