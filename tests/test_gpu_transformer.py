"""GPU parity: Transformer-encoder scorer (inference) through the C ABI vs golden vectors from the real reference."""
import numpy as np
import pytest
import torch

import recipes as R
from conftest import load_golden, js

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _load(m, w, dev):
    missing = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}, strict=False)
    assert not missing.unexpected_keys
    return m.eval().to(dev)


def test_transformer_small_goldens():
    from summarizer_amd.models.transformer import Transformer
    dev = torch.device("cuda:0")
    g = load_golden("transformer_small")
    for name, kw in js(g["meta"]).items():
        w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith(f"{name}/w/")}
        m = _load(Transformer(**kw), w, dev)
        for c in sorted(k.split("/")[-1] for k in g.files if k.startswith(f"{name}/x/")):
            with torch.no_grad():
                y = m(torch.from_numpy(g[f"{name}/x/{c}"].copy()).to(dev)).cpu().numpy()
            np.testing.assert_allclose(y, g[f"{name}/y/{c}"], atol=TOL, rtol=0, err_msg=f"{name} {c}")


def test_transformer_full_size_goldens_and_packed_batch():
    from oracle import transformer_np
    from summarizer_amd.models.transformer import Transformer
    dev = torch.device("cuda:0")
    g = load_golden("transformer_full")
    for ci in range(2):
        cfg = js(g[f"c{ci}/cfg"])
        w = R.transformer_weights(cfg["D"], cfg["layers"], cfg["wseed"])
        assert R.digest(w) == cfg["wdigest"]
        m = _load(Transformer(input_size=cfg["D"], encoder_layers=cfg["layers"], attention_heads=cfg["heads"]), w, dev)
        x = torch.from_numpy(R.features(cfg["T"], cfg["B"], cfg["D"], cfg["xseed"])).to(dev)
        with torch.no_grad():
            y = m(x).cpu().numpy()
        np.testing.assert_allclose(y, g[f"c{ci}/y"], atol=TOL, rtol=0, err_msg=str(cfg))
        if ci == 1:                                # grad-enabled call (eval mode: no dropout) = same scores through the training path
            np.testing.assert_allclose(m(x).detach().cpu().numpy(), g[f"c{ci}/y"], atol=TOL, rtol=0)
    # ragged packed batch vs the oracle (D=256, 4 heads)
    D, L, Hh = 256, 2, 4
    w = R.transformer_weights(D, L, 123)
    m = _load(Transformer(input_size=D, encoder_layers=L, attention_heads=Hh), w, dev)
    lens = [1, 2, 65, 130, 7]
    xs = [R.features(T, 1, D, 200 + i) - 0.1 for i, T in enumerate(lens)]
    with torch.no_grad():
        s = m.score_packed(torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev), lens).cpu().numpy()
    off = np.concatenate([[0], np.cumsum(lens)])
    for i, x in enumerate(xs):
        ref = transformer_np.transformer_forward(x, w, L, Hh)[:, 0, 0]
        np.testing.assert_allclose(s[off[i]:off[i + 1]], ref, atol=TOL, rtol=0, err_msg=f"video {i}")


@pytest.mark.parametrize("precision,tol_fp32", [("bf16x6", 1e-5), ("bf16x3", 1e-4)])
def test_transformer_plane_path_goldens_and_ragged_batch(precision, tol_fp32):
    """Inference in a split-bf16 arithmetic runs every projection of the stack on the plane GEMM (weights' planes cached per weight
    change, ReLU(linear1) as planes only): the REAL reference's full-size golden (D = 1024, 6 layers, 8 heads) at the usual gate, a
    ragged packed batch vs the oracle, and the distance to the exact-fp32 scores."""
    from oracle import transformer_np
    from summarizer_amd import _lib, kernels
    from summarizer_amd.models.transformer import Transformer
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = load_golden("transformer_full")
    cfg = js(g["c0/cfg"])
    w = R.transformer_weights(cfg["D"], cfg["layers"], cfg["wseed"])
    m = _load(Transformer(input_size=cfg["D"], encoder_layers=cfg["layers"], attention_heads=cfg["heads"]), w, dev)
    x = torch.from_numpy(R.features(cfg["T"], cfg["B"], cfg["D"], cfg["xseed"])).to(dev)
    sb = kernels.SeqBatch.get([cfg["T"]], dev)
    code = kernels.precision_code(precision)
    args = (cfg["D"], cfg["D"], cfg["heads"], cfg["layers"], sb.n_seq, sb.off_host_p, 0)
    assert lib.sumk_transformer_workspace_bytes_for(*args, code) > lib.sumk_transformer_workspace_bytes(*args)      # the plane path applies
    with torch.no_grad():
        y32 = m(x).cpu().numpy()
        m.precision = precision
        y = m(x).cpu().numpy()
        assert m._wpl is not None
        blk = m._wpl
        assert m(x).cpu().numpy().tobytes() == y.tobytes() and m._wpl is blk                    # repeatable; the block is cached ...
        w0 = m.k1.weight.clone()
        m.k1.weight.mul_(1.5)
        y_moved = m(x).cpu().numpy()
        m.k1.weight.copy_(w0)
        assert np.abs(y_moved - y).max() > 1e-4                                                    # ... and follows the weights
        np.testing.assert_array_equal(m(x).cpu().numpy(), y)
    np.testing.assert_allclose(y, g["c0/y"], atol=TOL, rtol=0)
    assert np.abs(y - y32).max() < tol_fp32, np.abs(y - y32).max()
    # ragged packed batches (D = 256, 2 layers): one-frame and two-frame videos among longer ones.  4 heads of 64 columns: the per-head
    # products stay on the in-loop kernels between plane GEMMs; 2 heads of 128 columns: the attention runs on planes too (attn_pw.hip's
    # multi-head form), first with every video inside 320 frames, then with one beyond it (the whole batch falls back)
    D, L = 256, 2
    w = R.transformer_weights(D, L, 123)
    for Hh, lens in ((4, [1, 2, 65, 130, 7, 300]), (2, [1, 2, 65, 130, 7, 300, 320, 64]), (2, [130, 7, 321])):
        m = _load(Transformer(input_size=D, encoder_layers=L, attention_heads=Hh), w, dev)
        m.precision = precision
        xs = [R.features(T, 1, D, 200 + i) - 0.1 for i, T in enumerate(lens)]
        with torch.no_grad():
            s = m.score_packed(torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev), lens).cpu().numpy()
        assert m._wpl is not None
        off = np.concatenate([[0], np.cumsum(lens)])
        for i, x in enumerate(xs):
            ref = transformer_np.transformer_forward(x, w, L, Hh)[:, 0, 0]
            np.testing.assert_allclose(s[off[i]:off[i + 1]], ref, atol=TOL, rtol=0, err_msg=f"heads {Hh} video {i}")


def test_transformer_plane_entry_points_check_their_arguments():
    """sumk_transformer_wplanes_build refuses a short or misaligned output block and ineligible widths; the forward refuses a workspace
    that lacks the plane buffers when it is handed weight planes -- statuses and messages, no crash, the library stays usable."""
    import ctypes as C
    from summarizer_amd import _lib, kernels
    from summarizer_amd._lib import SumkError
    from summarizer_amd.models.transformer import Transformer
    dev = torch.device("cuda:0")
    lib = _lib.load()
    D, L, Hh = 256, 2, 2
    m = Transformer(input_size=D, encoder_layers=L, attention_heads=Hh).to(dev).eval()
    p = {k: v.detach() for k, v in m.named_parameters()}
    layers, head = kernels._tf_structs(p, L, "weight", _lib.TfLayerWeights, _lib.TfHeadWeights)
    nb = lib.sumk_transformer_wplanes_bytes(D, D, L, 3)
    assert nb > 0
    buf = torch.empty(nb + 512, dtype=torch.uint8, device=dev)
    base = (buf.data_ptr() + 255) // 256 * 256
    args = (D, D, L, C.cast(layers, C.c_void_p), C.cast(C.pointer(head), C.c_void_p), 3)
    assert lib.sumk_transformer_wplanes_build(*args, C.c_void_p(base), nb - 1, None) == -1
    assert b"bytes" in lib.sumk_last_error()
    assert lib.sumk_transformer_wplanes_build(*args, C.c_void_p(base + 16), nb, None) == -1
    assert lib.sumk_transformer_wplanes_build(D, D, L, C.cast(layers, C.c_void_p), C.cast(C.pointer(head), C.c_void_p), 4, C.c_void_p(base), nb, None) == -1
    assert lib.sumk_transformer_wplanes_build(*args, C.c_void_p(base), nb, None) == 0
    # forward: weight planes given, workspace sized by the plain query -> refused with the name of the right query
    lens = [200, 150]
    sb = kernels.SeqBatch.get(lens, dev)
    x = torch.randn(sum(lens), D, device=dev)
    o = kernels._tf_opts(dict(layer_eps=1e-5, final_eps=1e-5, precision="bf16x6"))
    o.wplanes = base
    small = lib.sumk_transformer_workspace_bytes(D, D, Hh, L, sb.n_seq, sb.off_host_p, 0)
    assert lib.sumk_transformer_workspace_bytes_for(D, D, Hh, L, sb.n_seq, sb.off_host_p, 0, o.precision) > small
    ws = torch.empty(small, dtype=torch.uint8, device=dev)
    scores = torch.empty(sum(lens), device=dev)
    rc = lib.sumk_transformer_forward(x.data_ptr(), D, D, Hh, L, sb.n_seq, sb.off_host_p, sb.off_dev_p, C.cast(layers, C.c_void_p),
                                      C.cast(C.pointer(head), C.c_void_p), C.cast(C.pointer(o), C.c_void_p), None, None, scores.data_ptr(),
                                      ws.data_ptr(), small, 0, None)
    assert rc == -2 and b"sumk_transformer_workspace_bytes_for" in lib.sumk_last_error()
    m.precision = "bf16x6"
    with torch.no_grad():
        s = m.score_packed(x, lens)                 # the library is fine afterwards
    assert bool(torch.isfinite(s).all())
    with pytest.raises(SumkError):
        kernels.transformer_forward_packed(x[:-1], sb, p, L, Hh, D, dict(layer_eps=1e-5, final_eps=1e-5))


@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3"])
def test_transformer_plane_path_is_batch_independent(precision):
    """On the plane path a video's scores do not depend on what else is in the packed batch or where it sits in it (rows of the plane GEMMs,
    per-(video, head) attention strips and row kernels are independent; every accumulation order is fixed): bit-identical at BASELINE width."""
    from summarizer_amd.models.transformer import Transformer
    dev = torch.device("cuda:0")
    D = 1024
    torch.manual_seed(7)
    m = Transformer(input_size=D, encoder_layers=2, attention_heads=8).to(dev).eval()
    m.precision = precision
    lens = [300, 150, 320, 1, 201, 64]
    xs = [torch.from_numpy(R.features(T, 1, D, 900 + i)[:, 0, :]).to(dev) for i, T in enumerate(lens)]
    with torch.no_grad():
        full = torch.split(m.score_packed(torch.cat(xs), lens), lens)
        order = [2, 5, 0, 4]                                    # a sub-batch in another order: other row offsets, other tile positions
        part = torch.split(m.score_packed(torch.cat([xs[i] for i in order]), [lens[i] for i in order]), [lens[i] for i in order])
    assert m._wpl is not None
    for j, i in enumerate(order):
        assert torch.equal(part[j], full[i]), f"video {i}: {float((part[j] - full[i]).abs().max())}"
    assert all(bool(torch.isfinite(f).all()) and float(f.min()) >= 0 and float(f.max()) <= 1 for f in full)


@pytest.mark.parametrize("tag,kw", [("tf", dict(input_size=64, encoder_layers=2, attention_heads=4)),
                                    ("tf_res", dict(input_size=64, encoder_layers=1, attention_heads=8, more_residuals=True))])
def test_transformer_train_step_goldens(tag, kw):
    """MSE + Adam for 3 steps with every dropout forced to 0, vs the real reference (loss, all gradients, parameters)."""
    from summarizer_amd.models.transformer import Transformer
    from summarizer_amd import kernels
    dev = torch.device("cuda:0")
    g = load_golden("transformer_train")
    w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith(f"{tag}/w/")}
    m = Transformer(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    m = m.to(dev).train()
    m.dropout.p = 0.0
    for lyr in m.transformer_encoder.layers:
        lyr.dropout.p = 0.0                       # the HIP path reads layers[0].dropout.p as the layer dropout rate
    used = [(k, p) for k, p in m.named_parameters() if k in set(kernels.transformer_param_names(kw["encoder_layers"]))]
    x = torch.from_numpy(g["x"]).to(dev); tgt = torch.from_numpy(g["target"]).to(dev)
    opt = torch.optim.Adam([p for _, p in used], lr=5e-5, weight_decay=1e-5)
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
    for s in range(3):
        loss = torch.nn.functional.mse_loss(m(x.clone()), tgt)
        opt.zero_grad(); loss.backward()
        if s == 0:
            np.testing.assert_allclose(loss.item(), g[f"{tag}/loss0"], rtol=2e-5)
            for k, p in used:
                assert rel(p.grad.cpu().numpy(), g[f"{tag}/grad0/{k}"]) < 3e-4, (k, rel(p.grad.cpu().numpy(), g[f"{tag}/grad0/{k}"]))
        opt.step()
        if s in (0, 2):
            for k, p in used:
                # the key-bias slice of in_proj_bias has a mathematically ZERO gradient (softmax is invariant to it): what is
                # left is rounding noise, which Adam normalises to +-lr per step -- compare those entries at 3 * lr
                atol = 2e-4 if k.endswith("in_proj_bias") else 3e-6
                np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"{tag}/param{s+1}/{k}"], atol=atol, err_msg=f"{k} step {s+1}")


def test_transformer_trainer_with_dropout_runs():
    import random
    from summarizer_amd.models.transformer import TransformerTrainer
    from summarizer_amd.utils.datasets import synthetic_dataset
    from summarizer_amd.utils.hps import make_hps
    ds = synthetic_dataset(9, seed=3, D=128, t_range=(40, 90), n_users=5)
    keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))
    hps = make_hps(ds, [{"train_keys": keys[2:], "test_keys": keys[:2]}], epochs=3, test_every_epochs=2, lr=3e-4,
                   selection_algorithm="rank",
                   extra_params={"input_size": "128", "encoder_layers": "2", "attention_heads": "4", "batch_videos": "3"})
    torch.manual_seed(4); random.seed(4)
    tr = TransformerTrainer(hps, hps.splits_files[0]).reset()
    best = tr.train(0)                                  # dropout ON (0.1 in the layers, 0.5 before the head)
    losses = [v for _, v in hps.writer.scalars["synthetic/Fold_1/Train/Loss"]]
    assert all(np.isfinite(best)) and np.isfinite(losses).all() and losses[-1] < losses[0]
    # two forwards in training mode draw different masks; eval mode is deterministic
    x = torch.from_numpy(ds[keys[0]]["features"][...]).unsqueeze(1).cuda()
    with torch.no_grad():
        tr.model.train(); a = tr.model(x.clone()); b = tr.model(x.clone())
        tr.model.eval(); c = tr.model(x.clone()); d = tr.model(x.clone())
    assert torch.equal(c, d)
