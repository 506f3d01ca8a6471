"""CPU emulation: would a bf16x3 split (a = hi + lo, a*b ~= hi*hi + hi*lo + lo*hi, fp32 accumulate) keep VASNet scores
within the 1e-4 parity gate?  Compares against the float64 oracle on full-size (D=1024) cases."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import numpy as np, torch
import recipes as R
from oracle import vasnet_np

def split(x):
    hi = x.to(torch.bfloat16).to(torch.float32)
    lo = (x - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo

def mm3(a, b, mode):
    """a @ b with operands rounded as the chosen MFMA path would see them; fp32 accumulation (torch fp32 matmul of
    exactly-representable products is a faithful stand-in for the fp32 accumulator)."""
    if mode == "fp32":
        return a @ b
    ah, al = split(a); bh, bl = split(b)
    if mode == "bf16":
        return ah @ bh
    if mode == "bf16x3":
        return ah @ bh + (ah @ bl + al @ bh)
    raise ValueError(mode)

def vasnet(x, w, mode):
    t = lambda k: torch.from_numpy(w[k])
    X = torch.from_numpy(x[:, 0, :])
    D = X.shape[1]
    Q = mm3(X, t("Q.weight").T, mode); K = mm3(X, t("K.weight").T, mode); V = mm3(X, t("V.weight").T, mode)
    e = mm3(Q, K.T.contiguous(), mode) * (1.0 / np.sqrt(D))
    a = torch.softmax(e, dim=1)
    c = mm3(mm3(a, V, mode), t("attention_head_projection.weight").T, mode)
    y = torch.nn.functional.layer_norm(c + X, (D,), t("layer_norm.weight"), t("layer_norm.bias"), 1e-6)
    z = torch.relu(mm3(y, t("k1.weight").T, mode) + t("k1.bias"))
    z = torch.nn.functional.layer_norm(z, (D,), t("layer_norm.weight"), t("layer_norm.bias"), 1e-6)
    return torch.sigmoid(z @ t("k2.weight").T + t("k2.bias")).numpy()[:, 0]

torch.manual_seed(0)
for case, (T, wseed, xseed, init) in enumerate([(300, 7000, 7100, "recipe"), (163, 7001, 7101, "recipe"), (320, 1234, 55, "xavier")]):
    D = 1024
    if init == "recipe":
        w = R.vasnet_weights(D, wseed)
    else:
        from summarizer_amd.models.vasnet import VASNet
        torch.manual_seed(wseed); m = VASNet(input_size=D)
        w = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    x = R.features(T, 1, D, xseed)
    ref = vasnet_np.vasnet_forward(x, w, dtype=np.float64)[:, 0, 0]
    for mode in ("fp32", "bf16x3", "bf16"):
        y = vasnet(x, w, mode)
        print(f"case {case} T={T} init={init:7s} {mode:7s}: max|d| vs float64 oracle = {np.abs(y - ref).max():.2e}")
