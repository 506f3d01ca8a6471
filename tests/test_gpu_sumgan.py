"""SumGAN's forward-running LSTM stacks (eLSTM, cLSTM) on the HIP kernels vs goldens from the REAL reference modules
(tests/golden/make_golden_sumgan.py): outputs within 1e-4 (1e-5 typical), gradients of a fixed scalar loss w.r.t. the
input and every parameter within 3e-4 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def _run(g, tag, module, outputs_fn):
    dev = torch.device("cuda:0")
    module.load_state_dict({k.split("/w/")[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}/w/")})
    module = module.to(dev)
    x = torch.from_numpy(g[f"{tag}/x"]).to(dev).requires_grad_(True)
    outs = outputs_fn(module, x)
    loss = 0
    for i, o in enumerate(outs):
        ref = g[f"{tag}/y{i}"]
        assert tuple(o.shape) == ref.shape, (tag, i, o.shape, ref.shape)
        np.testing.assert_allclose(o.detach().cpu().numpy(), ref, atol=1e-4, rtol=0, err_msg=f"{tag} output {i}")
        loss = loss + (o * torch.from_numpy(g[f"{tag}/cw{i}"]).to(dev)).sum()
    loss.backward()
    assert _rel(x.grad.cpu().numpy(), g[f"{tag}/dx"]) < 3e-4, (tag, "dx", _rel(x.grad.cpu().numpy(), g[f"{tag}/dx"]))
    for k, p in module.named_parameters():
        r = _rel(p.grad.cpu().numpy(), g[f"{tag}/g/{k}"])
        assert r < 3e-4, (tag, k, r)
    # inference path (no autograd graph) gives the same outputs
    with torch.no_grad():
        outs2 = outputs_fn(module, x.detach())
    for o, o2 in zip(outs, outs2):
        np.testing.assert_allclose(o2.cpu().numpy(), o.detach().cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("T,B", [(37, 1), (20, 3)])
def test_elstm_vs_reference(T, B):
    from summarizer_amd.models.sumgan import eLSTM
    g = load_golden("sumgan_lstm")
    D, H, L = [int(v) for v in g["meta"]]
    _run(g, f"elstm_T{T}B{B}", eLSTM(D, H, L), lambda m, x: (lambda r: [r[0][0], r[0][1], r[1]])(m(x)))


@pytest.mark.parametrize("T,B", [(37, 1), (20, 3)])
def test_clstm_vs_reference(T, B):
    from summarizer_amd.models.sumgan import cLSTM
    g = load_golden("sumgan_lstm")
    D, H, L = [int(v) for v in g["meta"]]
    _run(g, f"clstm_T{T}B{B}", cLSTM(D, H, L), lambda m, x: list(m(x)))


def test_lstm_stack_initial_state_vs_torch():
    """Initial (h0, c0) and the gradients flowing into them, ragged packed batch, H not a multiple of 32: against torch's
    own nn.LSTM on the CPU, video by video."""
    from summarizer_amd import kernels
    from summarizer_amd.models._bilstm import lstm_stack
    dev = torch.device("cuda:0")
    D, H, L = 24, 40, 2
    torch.manual_seed(5)
    lstm = torch.nn.LSTM(D, H, num_layers=L)
    lens = [7, 1, 33, 12] + [3] * 30
    rng = np.random.default_rng(1)
    xs = [torch.from_numpy(rng.standard_normal((T, 1, D)).astype(np.float32)).requires_grad_(True) for T in lens]
    h0 = torch.from_numpy(rng.standard_normal((L, len(lens), H)).astype(np.float32) * 0.3).requires_grad_(True)
    c0 = torch.from_numpy(rng.standard_normal((L, len(lens), H)).astype(np.float32) * 0.3).requires_grad_(True)
    cw = [torch.from_numpy(rng.standard_normal((T, H)).astype(np.float32)) for T in lens]
    ch = torch.from_numpy(rng.standard_normal((L, len(lens), H)).astype(np.float32))
    cc = torch.from_numpy(rng.standard_normal((L, len(lens), H)).astype(np.float32))
    total = 0
    refs = []
    for i, x in enumerate(xs):
        o, (hn, cn) = lstm(x, (h0[:, i:i + 1].contiguous(), c0[:, i:i + 1].contiguous()))
        refs.append((o[:, 0].detach().numpy(), hn[:, 0].detach().numpy(), cn[:, 0].detach().numpy()))
        total = total + (o[:, 0] * cw[i]).sum() + (hn[:, 0] * ch[:, i]).sum() + (cn[:, 0] * cc[:, i]).sum()
    total.backward()
    ref_g = {k: p.grad.numpy().copy() for k, p in lstm.named_parameters()}
    ref_dx = [x.grad.numpy()[:, 0].copy() for x in xs]
    ref_dh0, ref_dc0 = h0.grad.numpy().copy(), c0.grad.numpy().copy()

    glstm = torch.nn.LSTM(D, H, num_layers=L)
    glstm.load_state_dict(lstm.state_dict()); glstm = glstm.to(dev)
    xp = torch.cat([x.detach()[:, 0] for x in xs]).to(dev).requires_grad_(True)
    gh0 = h0.detach().to(dev).requires_grad_(True); gc0 = c0.detach().to(dev).requires_grad_(True)
    sb = kernels.SeqBatch.get(lens, dev)
    out, hn, cn = lstm_stack(glstm, xp, sb, gh0, gc0)
    off = np.concatenate([[0], np.cumsum(lens)])
    for i in range(len(lens)):
        np.testing.assert_allclose(out[off[i]:off[i + 1]].detach().cpu().numpy(), refs[i][0], atol=1e-5)
        np.testing.assert_allclose(hn[:, i].detach().cpu().numpy(), refs[i][1], atol=1e-5)
        np.testing.assert_allclose(cn[:, i].detach().cpu().numpy(), refs[i][2], atol=1e-5)
    loss = (out * torch.cat(cw).to(dev)).sum() + (hn * ch.to(dev)).sum() + (cn * cc.to(dev)).sum()
    loss.backward()
    for k, p in glstm.named_parameters():
        assert _rel(p.grad.cpu().numpy(), ref_g[k]) < 3e-4, (k, _rel(p.grad.cpu().numpy(), ref_g[k]))
    gx = xp.grad.cpu().numpy()
    for i in range(len(lens)):
        assert _rel(gx[off[i]:off[i + 1]], ref_dx[i]) < 3e-4
    assert _rel(gh0.grad.cpu().numpy(), ref_dh0) < 3e-4 and _rel(gc0.grad.cpu().numpy(), ref_dc0) < 3e-4
