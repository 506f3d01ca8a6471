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


@pytest.mark.parametrize("T,B", [(37, 1), (20, 3)])
def test_dlstm_stepwise_decoder_vs_reference(T, B):
    """The reference decodes step by step in Python (sumgan.py:98-111); here it is one op.  Output, gradients of every
    parameter and of the initial state (h_0, c_0) against the real reference module."""
    from summarizer_amd.models.sumgan import dLSTM
    g = load_golden("sumgan_lstm")
    D, H, L = [int(v) for v in g["meta"]]
    tag = f"dlstm_T{T}B{B}"
    dev = torch.device("cuda:0")
    m = dLSTM(D, H, L)
    m.load_state_dict({k.split("/w/")[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}/w/")})
    m = m.to(dev)
    h0 = torch.from_numpy(g[f"{tag}/h0"]).to(dev).requires_grad_(True)
    c0 = torch.from_numpy(g[f"{tag}/c0"]).to(dev).requires_grad_(True)
    y = m(T, h0, c0)
    assert tuple(y.shape) == g[f"{tag}/y"].shape
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{tag}/y"], atol=1e-4, rtol=0)
    (y * torch.from_numpy(g[f"{tag}/cw"]).to(dev)).sum().backward()
    assert _rel(h0.grad.cpu().numpy(), g[f"{tag}/dh0"]) < 3e-4, _rel(h0.grad.cpu().numpy(), g[f"{tag}/dh0"])
    assert _rel(c0.grad.cpu().numpy(), g[f"{tag}/dc0"]) < 3e-4, _rel(c0.grad.cpu().numpy(), g[f"{tag}/dc0"])
    for k, p in m.named_parameters():
        r = _rel(p.grad.cpu().numpy(), g[f"{tag}/g/{k}"])
        assert r < 3e-4, (k, r)


def test_sumgan_container_state_dict_and_training_graph():
    """SumGAN / Summarizer / VAE / GAN: same state_dict keys as the reference container (recorded in the golden), the
    scorer is s_lstm, and one VAE + discriminator pass back-propagates finite gradients into EVERY parameter."""
    from summarizer_amd.models.sumgan import SumGAN
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = SumGAN(input_size=64, sLSTM_hidden_size=32, sLSTM_num_layers=2, edLSTM_hidden_size=48, edLSTM_num_layers=2,
               cLSTM_hidden_size=32, cLSTM_num_layers=2).to(dev)
    keys = set(m.state_dict().keys())
    for k in ("summarizer.s_lstm.lstm.weight_ih_l0_reverse", "summarizer.s_lstm.out.bias", "summarizer.vae.e_lstm.lstm.weight_hh_l1",
              "summarizer.vae.e_lstm.mu.weight", "summarizer.vae.e_lstm.logvar.bias", "summarizer.vae.d_lstm.lstm.bias_ih_l0",
              "summarizer.vae.d_lstm.recons.weight", "gan.c_lstm.lstm.weight_ih_l1", "gan.c_lstm.out.0.weight"):
        assert k in keys, k
    x = torch.randn(25, 1, 64, device=dev) * 0.5
    scores = m(x)
    assert scores.shape == (25, 1, 1) and bool(((scores > 0) & (scores < 1)).all())
    x_hat, (mu, logvar), s = m.summarizer(x)
    assert x_hat.shape == x.shape and mu.shape == (2, 1, 48) and s.shape == (25, 1, 1)
    p_real, h_real = m.gan(x)
    p_fake, h_fake = m.gan(x_hat)
    loss = (h_real - h_fake).pow(2).mean() + (-0.5 * (1 + logvar - mu.pow(2) - logvar.exp())).sum() + s.mean() \
        + torch.log(p_real + 1e-6).mean() + torch.log(1 - p_fake + 1e-6).mean()
    loss.backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, k


def test_sumgan_trainer_reproduces_the_reference_trainer_end_to_end():
    """G9: the REAL reference SumGANTrainer (CPU) was run in the build container (tests/golden/make_golden_e2e_sumgan.py):
    VAE pre-training epoch, then 2 epochs of selector+encoder / decoder / discriminator updates (three Adams, global
    gradient-norm clip over stale gradients included), supervised sparsity, input noise in epoch 0, rank selection -- with
    torch.randn_like / torch.rand replaced by the counter-based recipes.DetRandom.  Fed the same draws, the HIP trainer must
    start from the same weights, consume the same number of draws, follow the same six loss / probability curves, end at
    the same weights and report the same metrics."""
    import random
    import recipes as R
    from summarizer_amd.models.sumgan import SumGANTrainer
    from summarizer_amd.utils.datasets import synthetic_dataset
    from summarizer_amd.utils.hps import make_hps
    g = load_golden("e2e_sumgan")
    D, SEED, n, dseed, t0, t1, nu, n_draws = [int(v) for v in g["meta"]]
    ds = synthetic_dataset(n, seed=dseed, D=D, t_range=(t0, t1), n_users=nu)
    keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))
    ep = {"input_size": str(D), "sLSTM_hidden_size": "32", "edLSTM_hidden_size": "48", "cLSTM_hidden_size": "32",
          "pretrain_vae": "1", "epoch_noise": "1", "sup": True}
    hps = make_hps(ds, [{"train_keys": keys[3:], "test_keys": keys[:3]}], epochs=2, test_every_epochs=1, lr=1e-3,
                   selection_algorithm="rank", extra_params=ep)
    torch.manual_seed(SEED); random.seed(SEED)
    tr = SumGANTrainer(hps, hps.splits_files[0]).reset()
    for k, v in tr.model.state_dict().items():
        np.testing.assert_array_equal(v.detach().cpu().numpy(), g[f"w0/{k}"], err_msg=f"initial {k}")
    with R.DetRandom(SEED).patch() as det:
        best = tr.train(0)
        assert det.n == n_draws
    sc = hps.writer.scalars
    for t in ("Lse", "Ld", "Lc", "D_x", "D_x_hat", "D_x_hat_p"):
        got = [v for _, v in sc[f"synthetic/Fold_1/Train/{t}"]]
        np.testing.assert_allclose(got, g[t], rtol=2e-3, err_msg=t)
    worst = 0.0
    for k, v in tr.model.state_dict().items():
        d = float(np.abs(v.detach().cpu().numpy() - g[f"w1/{k}"]).max()); worst = max(worst, d)
        assert d < 2e-3, (k, d)                      # 42 Adam steps at lr 1e-3 move weights by up to 4e-2
    print("largest final-weight difference:", worst)
    np.testing.assert_allclose([v for _, v in sc["synthetic/Fold_1/Test/Correlation"]], g["corr"], atol=2e-2)
    f_avg = [v for _, v in sc["synthetic/Fold_1/Test/F-score_avg"]]; f_max = [v for _, v in sc["synthetic/Fold_1/Test/F-score_max"]]
    np.testing.assert_allclose(f_avg, g["f_avg"], atol=2e-2); np.testing.assert_allclose(f_max, g["f_max"], atol=2e-2)
    tr.model.eval()
    with torch.no_grad():
        for k in keys[:3]:
            s = tr.model(torch.from_numpy(ds[k]["features"][...]).unsqueeze(1).cuda()).squeeze().cpu().numpy()
            np.testing.assert_allclose(s, g[f"scores/{k}"], atol=5e-3)


def test_sumgan_batched_video_step_equals_the_pass_by_pass_one():
    """`train_video` batches the independent passes of each update; `train_video_sequential` runs the reference's 13 passes
    one by one.  Same draws -> same losses, same parameters after two video steps (up to summation order in the weight-
    gradient GEMMs)."""
    import random
    import recipes as R
    from summarizer_amd.models.sumgan import SumGANTrainer
    from summarizer_amd.utils.datasets import DictDataset
    from summarizer_amd.utils.hps import make_hps
    ep = {"input_size": "64", "sLSTM_hidden_size": "32", "edLSTM_hidden_size": "40", "cLSTM_hidden_size": "24", "sup": True}
    results = []
    for which in ("train_video", "train_video_sequential"):
        hps = make_hps(DictDataset({}), [{"train_keys": [], "test_keys": []}], epochs=1, lr=1e-3, extra_params=dict(ep))
        torch.manual_seed(9); random.seed(9)
        tr = SumGANTrainer(hps, hps.splits_files[0]).reset()
        tr.model.train(); tr.setup_optimizers()
        x = torch.from_numpy(R.features(45, 1, 64, 77)).cuda()
        y = torch.linspace(0, 1, 45, device="cuda").view(-1, 1, 1)
        with R.DetRandom(4).patch() as det:
            vals = [getattr(tr, which)(x, y, noisy=(i == 0)) for i in range(2)]
            draws = det.n
        results.append(([[float(v) for v in step[:6]] for step in vals], draws,
                        {k: v.detach().cpu().numpy().copy() for k, v in tr.model.state_dict().items()}))
    (la, da, wa), (lb, db, wb) = results
    assert da == db == 17                                   # 10 draws with noise, 7 without
    np.testing.assert_allclose(la, lb, rtol=1e-4)
    for k in wa:
        np.testing.assert_allclose(wa[k], wb[k], atol=2e-5, err_msg=k)
