"""GPU: the plane paths against the exact-fp32 path on random ragged batches whose sizes sit on the kernels' boundaries -- frame counts around
multiples of 32 / 64 / 192 / 256 (row tiles, plane pitch, strips), videos of 1 .. 321 frames (the attention on planes ends at 320), batches
below and above the XCD-map thresholds of the plane GEMM.  Scores must stay within the split arithmetics' distance from the fp32 scores and be
repeatable bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GATE = {"bf16x6": 2e-5, "bf16x3": 2e-4}           # distance to the fp32 scores (observed: 3e-6 / 3e-5)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _batches(rng, n):
    edge_T = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 319, 320]
    out = []
    for i in range(n):
        k = int(rng.integers(1, 24))
        lens = [int(rng.choice(edge_T)) if rng.random() < 0.6 else int(rng.integers(1, 321)) for _ in range(k)]
        if i % 5 == 0:
            lens[int(rng.integers(0, k))] = int(rng.choice([321, 400, 650]))          # one long video: the whole batch takes the fallback products
        target = int(rng.choice([0, 0, 256, 384, 576, 768, 3072, 3264]))               # pad the batch to a frame count on a tile boundary (+-1)
        if target and sum(lens) < target - 2:
            rest = target + int(rng.integers(-1, 2)) - sum(lens)
            while rest > 0:
                t = min(rest, int(rng.integers(1, 321))); lens.append(t); rest -= t
        out.append(lens)
    return out


@pytest.mark.parametrize("D,n_batches", [(256, 40), (1024, 12)])      # D = 1024: four / twelve column tiles, the XCD-aware tile maps engage from 16 row tiles
@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3"])
def test_vasnet_plane_paths_on_boundary_batches(dev, precision, D, n_batches):
    from summarizer_amd.models.vasnet import VASNet
    rng = np.random.default_rng(11 + D)
    torch.manual_seed(11)
    models = {fold: VASNet(input_size=D, fold_vo=fold).to(dev).eval() for fold in (False, True)}
    for lens in _batches(rng, n_batches):
        x = (torch.randn(sum(lens), D, device=dev).abs() * 0.5)
        for fold, m in models.items():
            with torch.no_grad():
                m.precision = "fp32"
                ref = m.score_packed(x, lens)
                m.precision = precision
                a = m.score_packed(x, lens)
                b = m.score_packed(x, lens)
            assert bool(torch.isfinite(a).all()), (lens, fold)
            assert torch.equal(a, b), (lens, fold)
            d = float((a - ref).abs().max())
            assert d < GATE[precision], (d, lens, fold)


@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3"])
def test_transformer_plane_path_on_boundary_batches(dev, precision):
    from summarizer_amd.models.transformer import Transformer
    rng = np.random.default_rng(12)
    D = 256
    torch.manual_seed(12)
    models = [Transformer(input_size=D, encoder_layers=2, attention_heads=h).to(dev).eval() for h in (2, 4)]
    for lens in _batches(rng, 24):
        x = (torch.randn(sum(lens), D, device=dev).abs() * 0.5)
        for m in models:
            with torch.no_grad():
                m.precision = "fp32"
                ref = m.score_packed(x, lens)
                m.precision = precision
                a = m.score_packed(x, lens)
                b = m.score_packed(x, lens)
            assert bool(torch.isfinite(a).all()), (lens, m.attention_heads)
            assert torch.equal(a, b), (lens, m.attention_heads)
            d = float((a - ref).abs().max())
            assert d < GATE[precision] * 2, (d, lens, m.attention_heads)


def test_dsn_projection_on_planes_on_boundary_batches(dev):
    """DSN in bf16x6: the input projection of both directions on the plane GEMM (1 024 packed frames and more), the recurrence in fp32 --
    against the all-fp32 path on batches around the eligibility threshold and the row-tile boundaries."""
    from summarizer_amd.models.dsn import DSN
    rng = np.random.default_rng(13)
    torch.manual_seed(13)
    m = DSN(1024, 256, 1).to(dev).eval()
    for total in (1000, 1023, 1024, 1025, 1151, 1152, 1153, 1344, 2049, 3071, 3072, 3073):
        lens, rest = [], total
        while rest > 0:
            t = min(rest, int(rng.choice([1, 2, 63, 64, 65, 300, int(rng.integers(1, 321))]))); lens.append(t); rest -= t
        x = (torch.randn(total, 1024, device=dev).abs() * 0.5)
        with torch.no_grad():
            m.precision = "fp32"
            ref = m.score_packed(x, lens)
            m.precision = "bf16x6"
            a = m.score_packed(x, lens)
            b = m.score_packed(x, lens)
        assert bool(torch.isfinite(a).all()), lens
        assert torch.equal(a, b), lens
        d = float((a - ref).abs().max())
        assert d < 2e-5, (d, total, lens)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_vasnet_training_step_on_boundary_batches(dev, precision):
    """Forward + backward of the packed training path (fused attention strips in bf16 for T <= 320, split-K weight gradients) on the same
    kind of batches: finite, repeatable bit for bit, and the bf16 gradients within mixed-precision distance of the fp32 ones."""
    from summarizer_amd.models.vasnet import VASNet
    rng = np.random.default_rng(14)
    D = 256
    torch.manual_seed(14)
    m = VASNet(input_size=D).to(dev).eval()                       # eval(): the training path without dropout

    def grads(x, lens, w, prec):
        m.precision = prec
        for p in m.parameters():
            p.grad = None
        s = m.score_packed(x, lens)
        (s * w).sum().backward()
        return [s.detach().clone()] + [p.grad.clone() for p in m.parameters()]
    for lens in _batches(rng, 16):
        x = (torch.randn(sum(lens), D, device=dev).abs() * 0.5)
        w = torch.rand(sum(lens), device=dev)
        a, b = grads(x, lens, w, precision), grads(x, lens, w, precision)
        for i, (u, v) in enumerate(zip(a, b)):
            assert bool(torch.isfinite(u).all()), (i, lens)
            assert torch.equal(u, v), (i, lens)
        if precision == "bf16":
            ref = grads(x, lens, w, "fp32")
            for i, (u, r) in enumerate(zip(a, ref)):
                scale = float(r.abs().max()) + 1e-6
                assert float((u - r).abs().max()) < 0.06 * scale, (i, float((u - r).abs().max()), scale, lens)
