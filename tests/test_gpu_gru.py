"""GPU parity of the reference's optional GRU cell, DSN(cell="gru") (dsn.py:28-47): outputs, loss and every gradient against
goldens from the REAL reference (tests/golden/gru_small.npz), a ragged packed batch against torch's nn.GRU autograd, and one
BASELINE-sized forward (D = 1024, H = 256, T = 300)."""
import numpy as np
import pytest
import torch

import recipes as R
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("L", [1, 2])
def test_gru_goldens_forward_and_gradients(L):
    from summarizer_amd.models.dsn import DSN
    dev = torch.device("cuda:0")
    g = load_golden("gru_small")
    tag = f"L{L}"
    m = DSN(input_size=64, hidden_size=16, num_layers=L, cell="gru")
    m.load_state_dict({k.split("/w/")[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}/w/")})
    m = m.to(dev)
    for c in sorted(k.split("/")[-1] for k in g.files if k.startswith(f"{tag}/x/")):
        with torch.no_grad():
            y = m(torch.from_numpy(g[f"{tag}/x/{c}"].copy()).to(dev)).cpu().numpy()
        np.testing.assert_allclose(y, g[f"{tag}/y/{c}"], atol=1e-5, rtol=0, err_msg=f"{tag} {c}")
    x = torch.from_numpy(R.features(37, 1, 64, 9371) - 0.2).to(dev).requires_grad_(True)
    loss = torch.nn.functional.mse_loss(m(x), torch.from_numpy(g[f"{tag}/target"]).to(dev))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{tag}/loss"], rtol=2e-5)
    for k, p in m.named_parameters():
        assert _rel(p.grad.cpu().numpy(), g[f"{tag}/grad/{k}"]) < 3e-4, (k, _rel(p.grad.cpu().numpy(), g[f"{tag}/grad/{k}"]))
    assert _rel(x.grad.cpu().numpy(), g[f"{tag}/dx"]) < 3e-4


def test_gru_ragged_packed_batch_vs_torch_port():
    from oracle import torch_port
    from summarizer_amd.models.dsn import DSN
    dev = torch.device("cuda:0")
    D, H, L, lens = 128, 40, 2, [50, 1, 33, 7, 64]
    torch.manual_seed(17)
    m = DSN(D, H, L, cell="gru")
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev)
    xs = [R.features(T, 1, D, 300 + i) - 0.2 for i, T in enumerate(lens)]
    xp = torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev).requires_grad_(True)
    cw = torch.from_numpy(np.random.default_rng(4).standard_normal(sum(lens)).astype(np.float32))
    s = m.score_packed(xp, lens)
    (s * cw.to(dev)).sum().backward()
    pt = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    gru = torch_port.make_gru({k: v.detach() for k, v in pt.items()}, "rnn.", D, H, L)
    off = np.concatenate([[0], np.cumsum(lens)])
    gx = []
    for i, x in enumerate(xs):
        xt = torch.from_numpy(x).clone().requires_grad_(True)
        y = torch_port.bigru_scores(xt, pt, "rnn.", "out.0.weight", "out.0.bias", gru)[:, 0, 0]
        np.testing.assert_allclose(s.detach().cpu().numpy()[off[i]:off[i + 1]], y.detach().numpy(), atol=1e-5, rtol=0)
        (y * cw[off[i]:off[i + 1]]).sum().backward()
        gx.append(xt.grad.numpy()[:, 0, :])
    ref = {f"rnn.{k}": v.grad.numpy() for k, v in gru.named_parameters()}
    ref["out.0.weight"], ref["out.0.bias"] = pt["out.0.weight"].grad.numpy(), pt["out.0.bias"].grad.numpy()
    for k, p in m.named_parameters():
        assert _rel(p.grad.cpu().numpy(), ref[k]) < 3e-4, (k, _rel(p.grad.cpu().numpy(), ref[k]))
    assert _rel(xp.grad.cpu().numpy(), np.concatenate(gx)) < 3e-4


def test_gru_baseline_size_forward_vs_torch_port():
    from oracle import torch_port
    from summarizer_amd.models.dsn import DSN
    dev = torch.device("cuda:0")
    D, H, T = 1024, 256, 300
    torch.manual_seed(23)
    m = DSN(D, H, 1, cell="gru")
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev).eval()
    x = R.features(T, 1, D, 777)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
        ref = torch_port.bigru_scores(torch.from_numpy(x), w, "rnn.", "out.0.weight", "out.0.bias", torch_port.make_gru(w, "rnn.", D, H, 1)).numpy()
    np.testing.assert_allclose(y, ref, atol=1e-4, rtol=0)
