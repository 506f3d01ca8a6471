"""debug: which backward roundings does the bf16 training step make?  HIP bf16 gradients of a few videos vs the emulated port with
(A) every backward operand rounded, (B) only the saved forward operands rounded (dy fp32), (C) fp32 backward on rounded-forward values"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipes as R
from oracle import torch_port
from summarizer_amd import kernels
from summarizer_amd.autograd import VasnetFunction
from summarizer_amd.models.vasnet import VASNet
dev = torch.device("cuda:0")
D = 1024
lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)][:int(sys.argv[1]) if len(sys.argv) > 1 else 50]
p_drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
w = R.vasnet_weights(D, 41)
xs = [R.features(T, 1, D, 5000 + i) - 0.1 for i, T in enumerate(lens)]
cw = np.random.default_rng(6).standard_normal(sum(lens)).astype(np.float32)
off = np.concatenate([[0], np.cumsum(lens)])
m = VASNet(input_size=D, precision="bf16"); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
xp = torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev).requires_grad_(True)
sb = kernels.SeqBatch.get(lens, dev)
opts = dict(scale=float(m.scale), eps=1e-6, ignore_self=False, aperture=None, dropout_p=p_drop, seed=777, precision="bf16")
names = [k for _, k in kernels.VASNET_FIELDS]
params = dict(m.named_parameters())
s = VasnetFunction.apply(xp, sb, opts, None, None, names, *[params[n] for n in names])
(s * torch.from_numpy(cw).to(dev)).sum().backward()
hip = {k: params[k].grad.cpu().numpy() for k in names}; hip["x"] = xp.grad.cpu().numpy(); hs = s.detach().cpu().numpy()
r16 = torch_port._r16
def l2(a, b):
    a = a.reshape(-1).astype(np.float64); b = b.reshape(-1).astype(np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
torch.set_num_threads(32)
for tag in ("A all rounded", "B dy fp32", "fp32 port"):
    if tag.startswith("B"):
        class L(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w_):
                xr, wr = r16(x), r16(w_); ctx.save_for_backward(xr, wr); return xr @ wr.t()
            @staticmethod
            def backward(ctx, dy):
                xr, wr = ctx.saved_tensors; return dy @ wr, dy.reshape(-1, dy.shape[-1]).t() @ xr.reshape(-1, xr.shape[-1])
        class B(torch.autograd.Function):
            @staticmethod
            def forward(ctx, a, b):
                ar, br = r16(a), r16(b); ctx.save_for_backward(ar, br); return torch.bmm(ar, br)
            @staticmethod
            def backward(ctx, dy):
                ar, br = ctx.saved_tensors; return torch.bmm(dy, br.transpose(1, 2)), torch.bmm(ar.transpose(1, 2), dy)
        oL, oB = torch_port._Linear16, torch_port._Bmm16
        torch_port._Linear16, torch_port._Bmm16 = L, B
    pt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in w.items()}
    sc, gx = [], []
    for i, x in enumerate(xs):
        xt = torch.from_numpy(x).clone().requires_grad_(True)
        y = torch_port.vasnet_scores(xt, pt, bf16_products=(tag != "fp32 port"))[:, 0, 0]
        (y * torch.from_numpy(cw[off[i]:off[i + 1]])).sum().backward()
        sc.append(y.detach().numpy()); gx.append(xt.grad.numpy()[:, 0, :])
    if tag.startswith("B"):
        torch_port._Linear16, torch_port._Bmm16 = oL, oB
    ref = {k: v.grad.numpy() for k, v in pt.items()}; ref["x"] = np.concatenate(gx)
    print(tag, "scores L2", f"{l2(hs, np.concatenate(sc)):.2e}", " ".join(f"{k.split('.')[0][:6]}.{k.split('.')[-1][:1]}={l2(hip[k], ref[k]):.1e}" for k in names + ["x"]))
