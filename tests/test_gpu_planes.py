"""GPU: the KB-plane format and the plane-aware wide GEMM (csrc/gemm_pw.hip) behind precision = bf16x6 / bf16x3 scoring.
Reference arithmetic: vasnet.py:114-131 multiplies fp32 tensors with torch.matmul; the planes are an exact (3 planes) or
2^-16-relative (2 planes) re-expression of the same fp32 operands, and the products are accumulated in fp32."""
import ctypes as C
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def decode_planes(buf, rows, K, n_planes):
    """KB planes -> (n_planes, rows, K) float64, by the layout formula of include/sumk.h."""
    raw = buf.cpu().numpy()
    pitch = (rows + 63) // 64 * 64
    body = raw[:pitch * K * n_planes * 2].view(np.uint16).reshape(K // 16, n_planes, 2, pitch, 8)
    out = np.zeros((n_planes, rows, K), dtype=np.float64)
    for p in range(n_planes):
        v = body[:, p]                                        # (kb, half, pitch, 8)
        v = np.transpose(v, (2, 0, 1, 3)).reshape(pitch, K)   # row, (kb, half, 8) = k
        out[p] = (v[:rows].astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return out, body


@pytest.mark.parametrize("n_planes", [3, 2])
def test_split_planes_layout_and_exactness(dev, n_planes):
    from summarizer_amd import kernels
    rng = np.random.default_rng(5)
    rows, K = 203, 96
    x = (rng.standard_normal((rows, K)) * np.exp(rng.uniform(-6, 6, (rows, K)))).astype(np.float32)
    x[3, 5] = 0.0; x[7, 9] = -0.0; x[11, 2] = 1e-30; x[12, 2] = 3e38
    xt = torch.from_numpy(x).to(dev)
    planes, body = decode_planes(kernels.split_planes(xt, n_planes), rows, K, n_planes)
    assert not body[:, :, :, rows:, :].any()                  # pad rows are zeros
    rec = planes.sum(0)
    if n_planes == 3:
        np.testing.assert_array_equal(rec.astype(np.float32), x)          # x1 + x2 + x3 == x exactly
    else:
        assert (np.abs(rec - x) <= 2.0 ** -16 * np.abs(x) + 1e-38).all()
    # plane 0 is bf16(x) round-to-nearest-even, as torch rounds it
    np.testing.assert_array_equal(planes[0].astype(np.float32), xt.to(torch.bfloat16).float().cpu().numpy())
    # a strided source (leading dimension > K)
    big = torch.from_numpy(np.concatenate([x, x[:, :32]], axis=1)).to(dev)
    p2, _ = decode_planes(kernels.split_planes(big[:, :K], n_planes), rows, K, n_planes)
    np.testing.assert_array_equal(p2, planes)


@pytest.mark.parametrize("shape", [(1000, 512, 256), (3001, 1024, 128), (577, 256, 1024), (1, 256, 128)])
@pytest.mark.parametrize("n_planes", [3, 2])
def test_gemm_planes_vs_float64_and_inloop_split(dev, shape, n_planes):
    """C = A B^T from planes: (a) within the error bound of the arithmetic against float64 (bf16x6: the fp32 bound the exact fp32 MFMA
    path is held to; bf16x3: 2^-15 |A| |B|^T), (b) BIT-IDENTICAL to the in-loop split kernels of precision bf16x6 / bf16x3 (same planes,
    same term order, same MFMA) -- the ragged last row tile, single-tile and XCD-mapped multi-round grids included."""
    from summarizer_amd import kernels, _lib
    lib = _lib.load()
    M, N, K = shape
    rng = np.random.default_rng(M + N + K + n_planes)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    at, bt = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    ap, bp = kernels.split_planes(at, n_planes), kernels.split_planes(bt, n_planes)
    c = kernels.gemm_planes(ap, M, bp, N, M, N, K, n_planes)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    bound = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
    err = np.abs(c.cpu().numpy() - ref)
    tol = (2.0 ** -21 if n_planes == 3 else 2.0 ** -15) * bound + 1e-30
    assert (err <= tol).all(), float((err / tol).max())
    old = torch.empty(M, N, dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.sumk_gemm_prec(0, at.data_ptr(), bt.data_ptr(), old.data_ptr(), M, N, K, 2 if n_planes == 3 else 1, st), "gemm_prec")
    assert torch.equal(c, old), float((c - old).abs().max())
    for variant in (1, 2):                                     # schedule variants of the probe: same arithmetic
        assert torch.equal(kernels.gemm_planes(ap, M, bp, N, M, N, K, n_planes, variant=variant), c)


def test_gemm_planes_rejects_ineligible_shapes(dev):
    from summarizer_amd import kernels
    from summarizer_amd._lib import SumkError
    x = torch.randn(300, 128, device=dev)
    w = torch.randn(200, 128, device=dev)            # N % 256 != 0
    xp, wp = kernels.split_planes(x, 3), kernels.split_planes(w, 3)
    with pytest.raises(SumkError):
        kernels.gemm_planes(xp, 300, wp, 200, 300, 200, 128, 3)
    with pytest.raises(SumkError):
        kernels.split_planes(torch.randn(10, 24, device=dev), 3)      # K % 16


# ------------------------------------------------------------------------------------------------ VASNet scoring on the plane path
def _batch(D, n, seed):
    sys_path_golden()
    import recipes as Rc
    lens = [int(t) for t in np.random.default_rng(seed).integers(150, 321, size=n)]
    x = np.concatenate([Rc.features(T, 1, D, 300 + i)[:, 0, :] for i, T in enumerate(lens)])
    return lens, x, Rc.vasnet_weights(D, seed + 1)


def sys_path_golden():
    import os, sys
    from conftest import GOLDEN
    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)


@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3"])
@pytest.mark.parametrize("D", [256, 1024])
def test_vasnet_plane_path_vs_inloop_split_and_port(dev, D, precision, fold):
    """VASNet.score_packed in bf16x6 / bf16x3 runs its three row-wise GEMMs on operand planes (vasnet.hip `pw`): scores against (a) the
    SAME arithmetic on the in-loop split kernels (a call without the plane pointers; the only difference is the summation order of the
    LayerNorm moments: 2e-6), (b) the fp32 oracle port of vasnet.py:101-148 at the 1e-4 gate."""
    from summarizer_amd import kernels
    from summarizer_amd.models.vasnet import VASNet
    from oracle import torch_port
    lens, x, w = _batch(D, 9, 21 + D)
    m = VASNet(input_size=D, precision=precision, fold_vo=fold).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
    xt = torch.from_numpy(x).to(dev)
    with torch.no_grad():
        got = m.score_packed(xt, lens)
    assert f"planes{kernels.PLANES_OF[precision]}" in xt._sumk_shadows and m._wpl is not None      # the plane path ran
    sb = kernels.SeqBatch.get(lens, dev)
    old, _ = kernels.vasnet_forward_packed(xt, sb, m._params(), m._opts(False), None, None, training=False, wvo=m._folded() if fold else None)
    assert float((got - old).abs().max()) < 2e-6, float((got - old).abs().max())
    ref = torch_port.vasnet_scores_packed(x, lens, w) if hasattr(torch_port, "vasnet_scores_packed") else None
    if ref is None:
        from oracle import vasnet_np
        off = np.concatenate([[0], np.cumsum(lens)])
        ref = np.concatenate([vasnet_np.vasnet_forward(x[off[i]:off[i + 1], None, :], w)[:, 0, 0] for i in range(len(lens))])
    d = float(np.abs(got.cpu().numpy() - ref).max())
    assert d < 1e-4, d
    with torch.no_grad():
        again = m.score_packed(xt, lens)
    assert torch.equal(again, got)                              # cached planes, same launch sequence: bit-repeatable


def test_weight_planes_follow_the_weights(dev):
    """The weight-plane block is rebuilt when the weights change -- through torch (tensor versions), through load_state_dict, and
    through the optimiser kernels of the C ABI (kernels.WEIGHTS_EPOCH) -- and reused otherwise."""
    from summarizer_amd import kernels
    from summarizer_amd.models.vasnet import VASNet
    D = 256
    lens, x, w = _batch(D, 4, 77)
    m = VASNet(input_size=D, precision="bf16x6").eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev)
    xt = torch.from_numpy(x).to(dev)
    sb = kernels.SeqBatch.get(lens, dev)

    def both():
        with torch.no_grad():
            got = m.score_packed(xt, lens)
        old, _ = kernels.vasnet_forward_packed(xt, sb, m._params(), m._opts(False), None, None, training=False)
        assert float((got - old).abs().max()) < 2e-6
        return got
    s0 = both()
    key0, blk0 = m._wpl_key, m._wpl.data_ptr()
    both()
    assert m._wpl_key == key0 and m._wpl.data_ptr() == blk0                      # reused
    with torch.no_grad():
        m.k1.weight.mul_(1.25)                                                  # torch-visible write
    s1 = both()
    assert m._wpl_key != key0 and not torch.equal(s0, s1)
    key1 = m._wpl_key
    g = torch.zeros_like(m.Q.weight); g.fill_(0.01)
    kernels.adam_step(m.Q.weight.data, g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-2)      # C-ABI write: WEIGHTS_EPOCH
    s2 = both()
    assert m._wpl_key != key1 and not torch.equal(s1, s2)
    x2 = xt.clone(); x2[5] += 1.0                                               # another input tensor: its own planes
    with torch.no_grad():
        s3 = m.score_packed(x2, lens)
    assert not torch.equal(s3, s2)
