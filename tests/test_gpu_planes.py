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
    if n_planes == 3:
        assert torch.equal(c, old), float((c - old).abs().max())
        for variant in (1, 2):                                 # schedule variants of the probe: same arithmetic
            assert torch.equal(kernels.gemm_planes(ap, M, bp, N, M, N, K, n_planes, variant=variant), c)
    else:
        # two planes run on the 16x16x32 MFMA shape (csrc/gemm_pw16.hip; K >= 160): a term is summed over 32 k before the next starts, so it
        # equals the 32x32x16 kernels to rounding; variant 32 keeps the 32x32x16 plane kernel, which IS bit-identical to the in-loop split
        c32 = kernels.gemm_planes(ap, M, bp, N, M, N, K, n_planes, variant=32)
        assert torch.equal(c32, old), float((c32 - old).abs().max())
        assert (np.abs(c.cpu().numpy() - old.cpu().numpy()) <= 2.0 ** -20 * bound + 1e-30).all()
        assert torch.equal(c, c32) == (K < 160)


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
    # (with T <= 320 the per-video attention runs on planes as well, csrc/attn_pw.hip: other tile shapes and summation orders of the same arithmetic)
    assert float((got - old).abs().max()) < (5e-6 if precision == "bf16x6" else 5e-5), float((got - old).abs().max())
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
        assert float((got - old).abs().max()) < 5e-6
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


# ------------------------------------------------------------------------------------------------ per-video attention on planes
@pytest.mark.parametrize("mask", [dict(), dict(ignore_self=1), dict(aperture=20)])
@pytest.mark.parametrize("n_planes", [3, 2])
@pytest.mark.parametrize("case", ["ragged8", "many_short"])
def test_attention_on_planes_vs_float64(dev, n_planes, mask, case):
    """csrc/attn_pw.hip through sumk_attn_planes: alpha = softmax(mask(Q K^T scale)) and context = alpha V per video (vasnet.py:118-131) from
    the planes of [Q | K | V], against float64 on the same fp32 inputs -- ragged lengths from 1 to 320 frames (1 to 5 strips, every key-tile
    count), both mask options.  The alpha planes are exactly the split of the fp32 alpha the kernel also writes (and zero from T up to the
    k32 step the context kernel ends on); the context planes sum to the fp32-grade product."""
    from summarizer_amd import kernels, _lib
    lib = _lib.load()
    # "ragged8": D = 256 (context launch on 64-query strips x all columns).  "many_short": 150 videos of 1 ... 140 frames at D = 512 -- the 128-query blocks x
    # half the columns of round 6, and a flat block list longer than one 64-video scan step of locate_block (three steps; ~190 / ~340 blocks over the 8 XCDs)
    D, lens = (256, [37, 64, 150, 320, 1, 200, 257, 96]) if case == "ragged8" else (512, [int(t) for t in np.random.default_rng(3).integers(1, 141, size=150)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    R = int(off[-1])
    rng = np.random.default_rng(11 + n_planes)
    qkv = rng.standard_normal((R, 3 * D)).astype(np.float32)
    qkv[:, :2 * D] *= 1.5
    scale = 0.06
    qt = torch.from_numpy(qkv).to(dev)
    qp = kernels.split_planes(qt, n_planes)
    ldes = [(t + 3) // 4 * 4 for t in lens]
    E = torch.zeros(int(sum(t * l for t, l in zip(lens, ldes))), dtype=torch.float32, device=dev)
    tmax = max(lens)
    AP = torch.zeros(lib.sumk_attn_planes_alpha_bytes(R, tmax, n_planes), dtype=torch.uint8, device=dev)
    CP = torch.zeros(lib.sumk_planes_bytes(R, D, n_planes), dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.sumk_attn_planes(qp.data_ptr(), R, D, n_planes, len(lens), _lib.host_i32(off), scale, mask.get("ignore_self", 0),
                                    mask.get("aperture", -1), E.data_ptr(), AP.data_ptr(), CP.data_ptr(), st), "sumk_attn_planes")
    Eh = E.cpu().numpy()
    K32 = (tmax + 31) // 32 * 32
    ap, _ = decode_planes(AP, R, K32, n_planes)
    cp, _ = decode_planes(CP, R, D, n_planes)
    tol_a, tol_c = (2e-6, 2e-5) if n_planes == 3 else (6e-5, 4e-4)
    if case == "many_short":        # (twice the contraction length and sharper rows -- alpha entries near 1: fp32's own rounding of exp is 2e-6 there)
        tol_a, tol_c = 2 * tol_a, 2 * tol_c
    eo = 0
    for s, T in enumerate(lens):
        r0 = int(off[s])
        q, k, v = (qkv[r0:r0 + T, i * D:(i + 1) * D].astype(np.float64) for i in range(3))
        lg = q @ k.T * np.float64(np.float32(scale))
        i, j = np.meshgrid(np.arange(T), np.arange(T), indexing="ij")
        if mask.get("ignore_self"):
            lg[i == j] = -np.inf
        if "aperture" in mask:
            lg[np.abs(i - j) > mask["aperture"]] = -np.inf
        if T == 1 and mask.get("ignore_self"):
            eo += T * ldes[s]
            continue                                          # a single self-masked frame: softmax of an empty row (NaN in the reference too)
        al = np.exp(lg - lg.max(1, keepdims=True)); al /= al.sum(1, keepdims=True)
        got = Eh[eo:eo + T * ldes[s]].reshape(T, ldes[s])[:, :T]
        eo += T * ldes[s]
        assert np.abs(got - al).max() < tol_a, (s, T, np.abs(got - al).max())
        # planes of alpha: plane 0 = bf16(alpha), the sum reproduces alpha (exactly with three planes), zeros behind T
        T32 = (T + 31) // 32 * 32
        rec = ap[:, r0:r0 + T, :T32].sum(0)
        if n_planes == 3:
            np.testing.assert_array_equal(rec[:, :T].astype(np.float32), got)
        else:
            assert (np.abs(rec[:, :T] - got) <= 2.0 ** -16 * got + 1e-38).all()
        assert not rec[:, T:].any()
        np.testing.assert_array_equal(ap[0, r0:r0 + T, :T].astype(np.float32), torch.from_numpy(got.copy()).to(torch.bfloat16).float().numpy())
        ctx = cp[:, r0:r0 + T].sum(0)
        ref = al @ v
        assert np.abs(ctx - ref).max() < tol_c, (s, T, np.abs(ctx - ref).max())


# ------------------------------------------------------------------------------------------------ BiLSTM scorers: input projection on planes
@pytest.mark.parametrize("kind", ["dsn", "slstm"])
def test_lstm_scorers_input_projection_on_planes(dev, kind):
    """DSN (dsn.py:38-47) and sLSTM (sumgan.py:23-46) with precision = bf16x6 / bf16x3: the input projection X W_ih^T + b_ih + b_hh of every
    layer runs on operand planes (csrc/gemm_pw.hip; x planes kept with the feature tensor, weight planes per weight change).  The layer output
    is BIT-IDENTICAL (bf16x6; bf16x3 runs the 16x16x32 MFMA shape: equal to rounding) to the same arithmetic on the in-loop split kernels (a call
    without the weight-plane block), and the scores stay within
    2e-6 (bf16x6: fp32-grade) / 5e-5 (bf16x3) of the exact-fp32 path; the planes follow a weight change."""
    from summarizer_amd import kernels
    sys_path_golden()
    import recipes as Rc
    D = 1024
    lens = [int(t) for t in np.random.default_rng(9).integers(150, 321, size=7)]
    x = torch.from_numpy(np.concatenate([Rc.features(T, 1, D, 700 + i)[:, 0, :] for i, T in enumerate(lens)])).to(dev)
    torch.manual_seed(3)
    if kind == "dsn":
        from summarizer_amd.models.dsn import DSN
        m = DSN(input_size=D).to(dev).eval(); prefix, H, layers = "rnn.", 256, 1
    else:
        from summarizer_amd.models.sumgan import sLSTM
        m = sLSTM(input_size=D).to(dev).eval(); prefix, H, layers = "lstm.", 1024, 2
    with torch.no_grad():
        ref = m.score_packed(x, lens)
        for prec, tol in (("bf16x6", 2e-6), ("bf16x3", 5e-5)):
            m.precision = prec
            got = m.score_packed(x, lens)
            assert m.__dict__["_sumk_wpl"][1][0] is not None and f"planes{kernels.PLANES_OF[prec]}" in x._sumk_shadows      # the plane path ran
            assert float((got - ref).abs().max()) < tol, (prec, float((got - ref).abs().max()))
            sb = kernels.SeqBatch.get(lens, dev)
            p = dict(m.named_parameters())
            wpl = m.__dict__["_sumk_wpl"][1]
            h_pl, _ = kernels.bilstm_layer_forward(x, sb, p, prefix, 0, H, precision=prec, wplanes=wpl[0], dataset_input=True)
            h_old, _ = kernels.bilstm_layer_forward(x, sb, p, prefix, 0, H, precision=prec)
            # (round 6: with planes at hand DSN's projection runs INSIDE the persistent recurrence -- csrc/lstm.hip, lstm_persist_proj_kernel --
            #  whose k order differs from the GEMM's: equal to rounding there; tests/test_gpu_lstm.py holds that path to the plane GEMM and the oracle)
            assert torch.equal(h_pl, h_old) if (prec == "bf16x6" and kind == "slstm") else float((h_pl - h_old).abs().max()) < 2e-6
        key0 = m.__dict__["_sumk_wpl"][0]
        dict(m.named_parameters())[prefix + "weight_ih_l0"].mul_(1.1)
        got2 = m.score_packed(x, lens)
        assert m.__dict__["_sumk_wpl"][0] != key0 and not torch.equal(got2, got)
        m.precision = "fp32"
        assert float((m.score_packed(x, lens) - got2).abs().max()) < 5e-5
    kernels.health_check()


@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3"])
def test_plane_path_at_stress_size_and_batching_independence(dev, precision):
    """(1) BASELINE config 5's shape on the plane path: ONE (T = 10 000, D = 2048) sequence -- T > 320, so the plane GEMMs run around the in-loop
    attention kernels -- against the exact-fp32 HIP path (itself held to the oracle port at this size by
    test_gpu_vasnet.py::test_vasnet_full_stress_size_vs_torch_port): scores within 1e-5 (bf16x6) / 1e-4 (bf16x3), logits within 5e-3 where they
    resolve.  (2) Batching independence of the whole plane path (GEMM row tiles mix videos, attention strips do not): videos scored together
    equal their separate scores bit for bit (each padded into a batch of the same plane-path eligibility)."""
    from summarizer_amd.models.vasnet import VASNet
    D, T = 2048, 10000
    torch.manual_seed(1234)
    m = VASNet(input_size=D).to(dev).eval()
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn(T, D, device=dev, generator=g) * 0.05
    with torch.no_grad():
        ref = m.score_packed(x, [T]).double()
        m.precision = precision
        got = m.score_packed(x, [T]).double()
    assert m._wpl is not None and f"planes{3 if precision == 'bf16x6' else 2}" in x._sumk_shadows
    tol = 1e-5 if precision == "bf16x6" else 1e-4
    assert float((got - ref).abs().max()) < tol, float((got - ref).abs().max())
    lg, lr = torch.log(got) - torch.log1p(-got), torch.log(ref) - torch.log1p(-ref)
    sel = lr.abs() <= 8.0
    assert int(sel.sum()) > T // 2 and float((lg - lr)[sel].abs().max()) < (5e-3 if precision == "bf16x6" else 5e-2)
    del x, ref, got
    # (2) at D = 1024 with T <= 320 (attention on planes): three videos together == the first two together + the third alone
    D2 = 1024
    sys_path_golden()
    import recipes as Rc
    lens = [300, 163, 320]
    xs = [torch.from_numpy(Rc.features(t, 1, D2, 900 + i)[:, 0, :]).to(dev) for i, t in enumerate(lens)]
    m2 = VASNet(input_size=D2, precision=precision).to(dev).eval()
    with torch.no_grad():
        allv = m2.score_packed(torch.cat(xs), lens)
        two = m2.score_packed(torch.cat(xs[:2]), lens[:2])
        one = m2.score_packed(xs[2], lens[2:])
    assert torch.equal(allv[:463], two) and torch.equal(allv[463:], one)


def test_plane_entry_points_argument_checks(dev):
    """Argument checks of the round-5 entry points: refused with SUMK_ERR_ARG and a message, nothing launched."""
    from summarizer_amd import kernels, _lib
    from summarizer_amd._lib import SumkError
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(300, 256, device=dev)
    with pytest.raises(SumkError, match="16-byte aligned"):
        buf = torch.empty(lib.sumk_planes_bytes(300, 256, 3) + 64, dtype=torch.uint8, device=dev)
        _lib.check(lib.sumk_split_planes(x.data_ptr(), 300, 256, 256, 3, buf.data_ptr() + 4, st), "split")
    with pytest.raises(SumkError, match="K % 16"):
        _lib.check(lib.sumk_split_planes(x.data_ptr(), 300, 250, 256, 3, buf.data_ptr(), st), "split")
    # attention on planes: a video longer than 320 frames, offsets that do not cover the rows
    D = 256
    qkv = torch.randn(700, 3 * D, device=dev)
    qp = kernels.split_planes(qkv, 3)
    ap = torch.empty(lib.sumk_attn_planes_alpha_bytes(700, 320, 3), dtype=torch.uint8, device=dev)
    off = np.array([0, 400, 700], dtype=np.int32)
    with pytest.raises(SumkError, match="not eligible"):
        _lib.check(lib.sumk_attn_planes(qp.data_ptr(), 700, D, 3, 2, _lib.host_i32(off), 0.1, 0, -1, None, ap.data_ptr(), None, st), "attn")
    off = np.array([0, 300, 600], dtype=np.int32)
    with pytest.raises(SumkError, match="bad arguments"):
        _lib.check(lib.sumk_attn_planes(qp.data_ptr(), 700, D, 3, 2, _lib.host_i32(off), 0.1, 0, -1, None, ap.data_ptr(), None, st), "attn")
    # weight-plane blocks: a buffer that is too small, a D the plane path does not take
    from summarizer_amd.models.vasnet import VASNet
    m = VASNet(input_size=256).to(dev).eval()
    w = _lib.VasnetWeights()
    for f, k in kernels.VASNET_FIELDS:
        setattr(w, f, dict(m.named_parameters())[k].data_ptr())
    small = torch.empty(4096, dtype=torch.uint8, device=dev)
    with pytest.raises(SumkError, match="needed"):
        _lib.check(lib.sumk_vasnet_wplanes_build(256, C.byref(w), None, 3, C.c_void_p((small.data_ptr() + 255) // 256 * 256), 1024, st), "wplanes")
    assert kernels.vasnet_wplanes({k: v.detach() for k, v in m.named_parameters()}, 200, 3) is None        # D % 256: no plane path, the caller keeps the in-loop kernels
    # a model whose D is not eligible simply scores on the in-loop kernels
    m2 = VASNet(input_size=192, precision="bf16x6").to(dev).eval()
    xs = torch.randn(400, 192, device=dev)
    with torch.no_grad():
        s = m2.score_packed(xs, [150, 250])
    assert bool(torch.isfinite(s).all()) and getattr(m2, "_wpl", None) is None


@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3"])
def test_plane_paths_read_nothing_stale_past_the_last_video(dev, precision):
    """The context kernel multiplies V rows up to 31 past the last video's end by alpha = 0; those rows (and the slack behind the last
    sub-array) must hold finite data whatever the workspace held before.  The scratch is poisoned with NaN bit patterns between two calls:
    same scores, bit for bit (a batch whose frame count leaves the last key block hanging past a multiple of 32 frames)."""
    from summarizer_amd import kernels
    from summarizer_amd.models.vasnet import VASNet
    from summarizer_amd.models.transformer import Transformer
    D = 256
    torch.manual_seed(3)
    for lens in ([200, 150], [300, 81], [129, 64, 190]):
        x = torch.randn(sum(lens), D, device=dev) * 0.3
        for m in (VASNet(input_size=D, precision=precision).to(dev).eval(), Transformer(input_size=D, encoder_layers=2, attention_heads=2).to(dev).eval()):
            m.precision = precision
            with torch.no_grad():
                m.score_packed(x, lens)                                   # (allocates the scratch)
                for buf in kernels._ws_cache.values():
                    buf.fill_(255)                                        # 0xFFFF = a bf16 NaN, 0xFFFFFFFF = an fp32 NaN
                a = m.score_packed(x, lens)
                for buf in kernels._ws_cache.values():
                    buf.fill_(255)
                b = m.score_packed(x, lens)
            assert m._wpl is not None
            assert bool(torch.isfinite(a).all()), (type(m).__name__, lens)
            assert torch.equal(a, b), (type(m).__name__, lens)



_LONG_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from summarizer_amd.models.vasnet import VASNet
D = 2048          # (the plane path of the row-wise GEMMs wants the large tiles: R D >= 8.4 M)
lens = [int(v) for v in sys.argv[2].split(",")]
torch.manual_seed(77)
g = torch.Generator().manual_seed(5)
x = (torch.randn(sum(lens), D, generator=g) * 0.3).to("cuda:0")
out = {}
for name, kw in (("plain", {}), ("aperture", dict(attention_aperture=40)), ("noself", dict(ignore_self=True))):
    torch.manual_seed(77)
    m = VASNet(input_size=D, **kw).to("cuda:0").eval()
    for prec in ("fp32", "bf16x6", "bf16x3"):
        m.precision = prec
        with torch.no_grad():
            out[f"{name}_{prec}"] = m.score_packed(x, lens).cpu().numpy()
            if prec != "fp32":      # the same videos in reverse order (row tiles of the row-wise GEMMs then mix other neighbours)
                off = np.concatenate([[0], np.cumsum(lens)])
                xr = torch.cat([x[off[i]:off[i + 1]] for i in reversed(range(len(lens)))])
                out[f"{name}_{prec}_reversed"] = m.score_packed(xr, lens[::-1]).cpu().numpy()
np.savez(sys.argv[1], **out)
'''


def test_long_videos_on_the_plane_gemm(tmp_path):
    """Round 6: in bf16x6 / bf16x3 a batch whose videos all have T >= 1536 runs its per-video (T x T) products on the plane GEMM itself
    (csrc/vasnet.hip `pw_long`: planes of Q_s, K_s, V_s^T per video, raw logits, softmax_planes_kernel -> planes of alpha, context straight
    into the batch's CTX planes) instead of the in-loop grouped kernels between plane GEMMs (SUMK_PW_LONG=0).  Both are the same arithmetic
    in other tile shapes and summation orders: scores agree within 5e-6 (three planes) / 5e-5 (two) and stay within the same bounds of
    exact fp32 -- default attention, a banded mask (`attention_aperture`: the tril * triu == 0 rule) and `ignore_self` -- on ragged
    lengths (1 600, 2 049, 1 537: none a multiple of the 192 / 256 / 32 tile and pad sizes); the batch in reverse order gives every video the
    same scores bit for bit (one set of launches per video)."""
    import os, subprocess, sys
    lens = [1600, 2049, 1537]
    out = {}
    for tag, env in (("inloop", {"SUMK_PW_LONG": "0"}), ("long", {})):
        f = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", _LONG_CHILD, str(f), ",".join(map(str, lens))], env=dict(os.environ, **env),
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = dict(np.load(f))
    ran_other_kernels = False
    for name in ("plain", "aperture", "noself"):
        ref = out["long"][f"{name}_fp32"]
        assert np.array_equal(ref, out["inloop"][f"{name}_fp32"])
        for prec, tol in (("bf16x6", 5e-6), ("bf16x3", 5e-5)):
            a, b = out["long"][f"{name}_{prec}"], out["inloop"][f"{name}_{prec}"]
            assert np.isfinite(a).all()
            assert np.abs(a - b).max() < tol, (name, prec, float(np.abs(a - b).max()))
            assert np.abs(a - ref).max() < 2 * tol, (name, prec, float(np.abs(a - ref).max()))
            rev, off, offr = out["long"][f"{name}_{prec}_reversed"], np.concatenate([[0], np.cumsum(lens)]), np.concatenate([[0], np.cumsum(lens[::-1])])
            for i in range(len(lens)):
                jr = len(lens) - 1 - i
                assert np.array_equal(rev[offr[jr]:offr[jr + 1]], a[off[i]:off[i + 1]]), (name, prec, i)
            ran_other_kernels |= not np.array_equal(a, b)
    assert ran_other_kernels


_RECUT_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from summarizer_amd.models.vasnet import VASNet
from summarizer_amd.models.transformer import Transformer
D = 1024
lens = [int(v) for v in sys.argv[2].split(",")]
g = torch.Generator().manual_seed(9)
x = (torch.randn(sum(lens), D, generator=g) * 0.3).to("cuda:0")
out = {}
for name, kw in (("plain", {}), ("folded", dict(fold_vo=True)), ("aperture", dict(attention_aperture=30))):
    torch.manual_seed(3)
    m = VASNet(input_size=D, **kw).to("cuda:0").eval()
    for prec in ("bf16x6", "bf16x3"):
        m.precision = prec
        with torch.no_grad():
            out[f"vasnet_{name}_{prec}"] = m.score_packed(x, lens).cpu().numpy()
torch.manual_seed(4)
t = Transformer(input_size=D).to("cuda:0").eval()
for prec in ("bf16x6", "bf16x3"):
    t.precision = prec
    with torch.no_grad():
        out[f"transformer_{prec}"] = t.score_packed(x, lens).cpu().numpy()
np.savez(sys.argv[1], **out)
'''


def test_attention_recut_switches_do_not_change_a_bit(tmp_path):
    """Round 6's re-cut of the attention launches on planes is a change of WHO computes WHAT, not of the arithmetic: per accumulator the plane products and the
    key order are the same.  `SUMK_ATTN_WIDE=0` (context launch on 64-query strips x all columns instead of 128-query blocks x half the columns) and
    `SUMK_ATTN_NT=0 / 1` (cache policy of the logits launch's K / Q loads) must reproduce the default's scores BIT FOR BIT -- VASNet plain / folded (the context
    launch adds the residual and emits the LayerNorm moments) / banded mask and the Transformer scorer (multi-head forms), both plane counts, on a ragged batch
    (1 ... 320 frames; enough rows for the plane path)."""
    import os, subprocess, sys
    lens = [320, 1, 129, 64, 255, 300, 2, 200] + [int(t) for t in np.random.default_rng(8).integers(100, 321, size=42)]
    out = {}
    for tag, env in (("default", {}), ("strips64", {"SUMK_ATTN_WIDE": "0"}), ("nt0", {"SUMK_ATTN_NT": "0"}), ("nt1", {"SUMK_ATTN_NT": "1"})):
        f = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", _RECUT_CHILD, str(f), ",".join(map(str, lens))], env=dict(os.environ, **env),
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = dict(np.load(f))
    assert len(out["default"]) == 8
    for k, ref in out["default"].items():
        assert np.isfinite(ref).all(), k
        for tag in ("strips64", "nt0", "nt1"):
            assert np.array_equal(out[tag][k], ref), (tag, k, float(np.abs(out[tag][k] - ref).max()))
