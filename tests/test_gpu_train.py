"""GPU parity of the TRAINING path: HIP backward kernels + Adam vs (a) golden gradients / parameters produced by the
real reference (tests/golden/train_small.npz: MSE, Adam lr 5e-5 wd 1e-5, vasnet.py:176-212 with dropout off) and
(b) autograd through the stock-PyTorch port of the oracle, including training-mode dropout with the deterministic
keep-masks (recipes.dropout_keep mirrors the kernel's hash bit for bit)."""
import numpy as np
import pytest
import torch

import recipes as R
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("precision,gtol", [("fp32", 2e-4), ("bf16x3", 1e-3), ("bf16x6", 2e-4)])
@pytest.mark.parametrize("tag,kw", [("vasnet", dict()), ("vasnet_loc", dict(attention_aperture=4, ignore_self=True))])
def test_vasnet_train_step_goldens(dev, tag, kw, precision, gtol):
    from summarizer_amd.models.vasnet import VASNet
    g = load_golden("train_small")
    w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith(f"{tag}/w/")}
    m = VASNet(input_size=64, precision=precision, **kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m = m.to(dev).eval()            # eval(): dropout off, exactly how the golden was generated
    x = torch.from_numpy(g["vasnet/x"]).to(dev); tgt = torch.from_numpy(g["vasnet/target"]).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=5e-5, weight_decay=1e-5)
    for s in range(3):
        scores = m(x)
        loss = torch.nn.functional.mse_loss(scores, tgt)
        opt.zero_grad(); loss.backward()
        if s == 0:
            np.testing.assert_allclose(loss.item(), g[f"{tag}/loss0"], rtol=1e-4 if precision == "bf16x3" else 2e-5)
            for k, p in m.named_parameters():
                ref = g[f"{tag}/grad0/{k}"]
                assert _rel(p.grad.cpu().numpy(), ref) < gtol, (k, _rel(p.grad.cpu().numpy(), ref))
        opt.step()
        if s in (0, 2):
            for k, p in m.named_parameters():
                np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"{tag}/param{s+1}/{k}"], atol=5e-6 if precision == "bf16x3" else 2e-6, err_msg=f"{k} step {s+1}")


def test_vasnet_grads_vs_torch_port_ragged_batch_with_dropout(dev):
    from oracle import torch_port
    from summarizer_amd import kernels
    from summarizer_amd.models.vasnet import VASNet
    D, lens, p, seed = 128, [70, 1, 33, 129], 0.5, 1234567
    w = R.vasnet_weights(D, 9)
    m = VASNet(input_size=D); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
    xs = [R.features(T, 1, D, 60 + i) - 0.15 for i, T in enumerate(lens)]
    xp = torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev).requires_grad_(True)
    sb = kernels.SeqBatch.get(lens, dev)
    opts = dict(scale=float(m.scale), eps=1e-6, ignore_self=False, aperture=None, dropout_p=p, seed=seed)
    from summarizer_amd.autograd import VasnetFunction
    names = [k for _, k in kernels.VASNET_FIELDS]
    params = dict(m.named_parameters())
    s = VasnetFunction.apply(xp, sb, opts, None, None, names, *[params[n] for n in names])
    cw = torch.from_numpy(np.random.default_rng(2).standard_normal(sum(lens)).astype(np.float32)).to(dev)
    (s * cw).sum().backward()
    # reference: per video through the torch port with the SAME masks
    masks = R.vasnet_drop_masks(seed, p, lens, D)
    pt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in w.items()}
    off = np.concatenate([[0], np.cumsum(lens)])
    total = 0
    xrefs = []
    for i, x in enumerate(xs):
        xt = torch.from_numpy(x).clone().requires_grad_(True); xrefs.append(xt)
        dm = tuple(torch.from_numpy(mm).unsqueeze(0) for mm in masks[i])
        y = torch_port.vasnet_scores(xt, pt, drop_masks=dm)[:, 0, 0]
        np.testing.assert_allclose(s.detach().cpu().numpy()[off[i]:off[i + 1]], y.detach().numpy(), atol=1e-4)
        total = total + (y * cw.cpu()[off[i]:off[i + 1]]).sum()
    total.backward()
    for k in names:
        got, ref = params[k].grad.cpu().numpy(), pt[k].grad.numpy()
        assert _rel(got, ref) < 3e-4, (k, _rel(got, ref))
    gx = xp.grad.cpu().numpy()
    for i, xt in enumerate(xrefs):
        assert _rel(gx[off[i]:off[i + 1]], xt.grad.numpy()[:, 0, :]) < 3e-4


def test_dropout_mask_statistics_and_determinism(dev):
    keep = R.dropout_keep(42, 1, np.arange(1 << 20, dtype=np.uint64), 0.5)
    assert abs(keep.mean() - 0.5) < 2e-3
    assert not np.array_equal(keep, R.dropout_keep(43, 1, np.arange(1 << 20, dtype=np.uint64), 0.5))
    assert abs(R.dropout_keep(42, 2, np.arange(1 << 18, dtype=np.uint64), 0.1).mean() - 0.9) < 3e-3


def test_adam_and_sumsq_vs_torch(dev):
    from summarizer_amd import kernels
    rng = np.random.default_rng(0)
    n = 100003
    p0 = rng.standard_normal(n).astype(np.float32); g = [rng.standard_normal(n).astype(np.float32) for _ in range(3)]
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([pt], lr=5e-5, weight_decay=1e-5)
    p = torch.from_numpy(p0.copy()).to(dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step, gi in enumerate(g, 1):
        pt.grad = torch.from_numpy(gi.copy()); opt.step()
        kernels.adam_step(p, torch.from_numpy(gi).to(dev), m, v, step, 5e-5, weight_decay=1e-5)
    np.testing.assert_allclose(p.cpu().numpy(), pt.detach().numpy(), atol=1e-6, rtol=1e-6)
    ss = kernels.sumsq(torch.from_numpy(g[0]).to(dev))
    np.testing.assert_allclose(ss.item(), float((g[0].astype(np.float64) ** 2).sum()), rtol=1e-5)


def test_dsn_train_step_goldens(dev):
    """DSN (BiLSTM 64->2x16) MSE + Adam for 3 steps vs the real reference (loss, all gradients, parameters)."""
    from summarizer_amd.models.dsn import DSN
    g = load_golden("train_small")
    w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith("dsn/w/")}
    m = DSN(64, 16, 1); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
    x = torch.from_numpy(g["vasnet/x"]).to(dev); tgt = torch.from_numpy(g["vasnet/target"]).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=5e-5, weight_decay=1e-5)
    for s in range(3):
        loss = torch.nn.functional.mse_loss(m(x), tgt)
        opt.zero_grad(); loss.backward()
        if s == 0:
            np.testing.assert_allclose(loss.item(), g["dsn/loss0"], rtol=2e-5)
            for k, p in m.named_parameters():
                assert _rel(p.grad.cpu().numpy(), g[f"dsn/grad0/{k}"]) < 2e-4, k
        opt.step()
        if s in (0, 2):
            for k, p in m.named_parameters():
                np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"dsn/param{s+1}/{k}"], atol=2e-6, err_msg=f"{k} step {s+1}")


def test_dsn_bce_grads_golden(dev):
    from summarizer_amd.models.dsn import DSN
    g = load_golden("train_small")
    w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith("dsn/w/")}
    m = DSN(64, 16, 1); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
    x = torch.from_numpy(g["vasnet/x"]).to(dev); tgt = torch.from_numpy(g["vasnet/target"]).to(dev)
    loss = torch.nn.functional.binary_cross_entropy(m(x), tgt)          # dsn.py:76,117-119 (sup extension)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["dsn_bce/loss0"], rtol=2e-5)
    for k, p in m.named_parameters():
        assert _rel(p.grad.cpu().numpy(), g[f"dsn_bce/grad0/{k}"]) < 2e-4, k


@pytest.mark.parametrize("precision,gtol", [("fp32", 3e-4), ("bf16x3", 1e-3)])
@pytest.mark.parametrize("kind,D,H,L,lens", [("dsn", 128, 40, 1, [50, 1, 33, 7] + [4] * 30), ("slstm", 64, 32, 2, [37, 90, 2]),
                                              ("slstm", 64, 264, 2, [21, 40, 2, 1] + [3] * 64),   # H > 256: wide persistent forward, > 64 videos
                                              ("slstm", 64, 264, 2, [21, 40, 2, 1, 9]),            # H > 256, few videos: small-batch mat-vec BPTT
                                              ("slstm", 64, 384, 2, [21, 40, 2, 1] + [3] * 64 + [17] * 63),   # wide persistent BPTT: KG 12 x NG 2, 3 groups (ragged, last partial)
                                              ("slstm", 64, 512, 1, [9, 30, 1, 14, 22, 5, 17, 3, 30, 11, 8, 2]),      # wide persistent BPTT, one MFMA tile (9 < videos <= 32), KG 16 x NG 2
                                              ("slstm", 128, 1024, 2, [int(v) for v in np.random.default_rng(9).integers(1, 70, 50)])])   # wide persistent BPTT at sLSTM's H: 32 x 4 members per direction
def test_bilstm_grads_vs_torch_port_ragged_batch(dev, kind, D, H, L, lens, precision, gtol):
    from oracle import torch_port
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.models.sumgan import sLSTM
    pre, hw, hb = ("rnn.", "out.0.weight", "out.0.bias") if kind == "dsn" else ("lstm.", "out.weight", "out.bias")
    w = R.lstm_weights(pre, D, H, L, 21, hw[:-6])
    m = DSN(D, H, L) if kind == "dsn" else sLSTM(D, H, L)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
    m.precision = precision
    xs = [R.features(T, 1, D, 80 + i) - 0.2 for i, T in enumerate(lens)]
    xp = torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev).requires_grad_(True)
    s = m.score_packed(xp, lens)
    cw = torch.from_numpy(np.random.default_rng(4).standard_normal(sum(lens)).astype(np.float32)).to(dev)
    (s * cw).sum().backward()
    pt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in w.items()}
    lstm = torch_port.make_lstm({k: v.detach() for k, v in pt.items()}, pre, D, H, L)
    off = np.concatenate([[0], np.cumsum(lens)])
    total, xrefs = 0, []
    for i, x in enumerate(xs):
        xt = torch.from_numpy(x).clone().requires_grad_(True); xrefs.append(xt)
        y = torch_port.bilstm_scores(xt, pt, pre, hw, hb, D, H, L, lstm=lstm)[:, 0, 0]
        np.testing.assert_allclose(s.detach().cpu().numpy()[off[i]:off[i + 1]], y.detach().numpy(), atol=1e-4)
        total = total + (y * cw.cpu()[off[i]:off[i + 1]]).sum()
    total.backward()
    ref_grads = {f"{pre}{k}": v.grad.numpy() for k, v in lstm.named_parameters()}
    ref_grads[hw] = pt[hw].grad.numpy(); ref_grads[hb] = pt[hb].grad.numpy()
    for k, p in m.named_parameters():
        assert _rel(p.grad.cpu().numpy(), ref_grads[k]) < gtol, (k, _rel(p.grad.cpu().numpy(), ref_grads[k]))
    gx = xp.grad.cpu().numpy()
    for i, xt in enumerate(xrefs):
        assert _rel(gx[off[i]:off[i + 1]], xt.grad.numpy()[:, 0, :]) < gtol


def test_lstm_launch_chain_path_still_matches(dev):
    """H > 256 (sLSTM) takes the launch-per-step kernels; force that path for the small golden cases too
    (SUMK_LSTM_PERSIST=0 is read once per process, hence the subprocess)."""
    import os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, SUMK_LSTM_PERSIST="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                        os.path.join(ROOT, "tests", "test_gpu_train.py"), "-k", "(dsn_train_step or bilstm_grads) and not 1024",   # (the H = 1024 case costs 2 min of CPU reference per process)
                        os.path.join(ROOT, "tests", "test_gpu_lstm.py")], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
