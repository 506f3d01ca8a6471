"""GPU: no kernel reads memory it (or a kernel before it in the same call) did not write.  Every scratch / output tensor the package
allocates on the device (`torch.empty`, `torch.empty_like`, the grow-only inference workspace) is filled with 0xFF bytes -- NaN as fp32 and
as bf16, -1 as an integer -- before it is handed out; scores and gradients must come out bit-identical to an unpoisoned run.  (Round 5: the
[Q | K | V] planes of the plane path left rows between the next multiple of 32 frames and the row pitch unwritten, and a context strip
multiplied them by alpha = 0 -- NaN for some frame counts only.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
LENS_LONG = [330, 64, 100]                   # one video above 320 frames: the plane paths keep the per-video products on the in-loop kernels
LENS = [1, 2, 65, 130, 22, 300, 150]         # 670 frames: the last video's last key block (keys 144..159) ends 8 rows past the next multiple of 32 frames


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class Poison:
    """Context manager: device allocations without initialisation come back as all-ones bytes."""
    def __init__(self, monkeypatch):
        self.mp = monkeypatch

    def __enter__(self):
        from summarizer_amd import kernels
        empty, empty_like, ws = torch.empty, torch.empty_like, kernels.workspace

        def fill(t):
            if isinstance(t, torch.Tensor) and t.is_cuda and t.numel():
                if t.is_contiguous():
                    t.reshape(-1).view(torch.uint8).fill_(255)
            return t
        self.mp.setattr(torch, "empty", lambda *a, **k: fill(empty(*a, **k)))
        self.mp.setattr(torch, "empty_like", lambda *a, **k: fill(empty_like(*a, **k)))
        self.mp.setattr(kernels, "workspace", lambda *a, **k: fill(ws(*a, **k)))
        return self

    def __exit__(self, *exc):
        self.mp.undo()
        return False


def _both(monkeypatch, fn):
    clean = fn()
    with Poison(monkeypatch):
        dirty = fn()
    return clean, dirty


def _same(clean, dirty, what):
    clean = clean if isinstance(clean, (list, tuple)) else [clean]
    dirty = dirty if isinstance(dirty, (list, tuple)) else [dirty]
    assert len(clean) == len(dirty)
    for i, (a, b) in enumerate(zip(clean, dirty)):
        assert bool(torch.isfinite(b).all()), f"{what}[{i}]: not finite under poisoned scratch"
        assert torch.equal(a, b), f"{what}[{i}]: differs by {float((a - b).abs().max())} under poisoned scratch"


def _x(D, dev, seed=0, lens=None):
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    return (torch.randn(sum(lens or LENS), D, generator=g).abs() * 0.5).to(dev)


@pytest.mark.parametrize("precision", ["fp32", "bf16x6", "bf16x3"])
@pytest.mark.parametrize("D", [64, 256])
def test_vasnet_scoring(dev, monkeypatch, D, precision):
    from summarizer_amd.models.vasnet import VASNet
    torch.manual_seed(1)
    for lens in (LENS, LENS_LONG):
        x = _x(D, dev, 0, lens)
        for fold in (False, True):
            m = VASNet(input_size=D, precision=precision, fold_vo=fold).to(dev).eval()

            def run():
                from summarizer_amd import kernels
                kernels.drop_shadows(x)                  # the planes / bf16 copies of x and the weight-plane block are rebuilt in both runs
                m._wpl = None
                with torch.no_grad():
                    return m.score_packed(x, lens).clone()
            _same(*_both(monkeypatch, run), f"VASNet D={D} {precision} fold={fold} lens={lens}")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_vasnet_gradients(dev, monkeypatch, precision):
    from summarizer_amd.models.vasnet import VASNet
    torch.manual_seed(2)
    D = 256
    x = _x(D, dev, 1)
    m = VASNet(input_size=D, precision=precision).to(dev).eval()          # eval(): the training path without dropout
    w = torch.rand(sum(LENS), device=dev)

    def run():
        from summarizer_amd import kernels
        kernels.drop_shadows(x)
        for p in m.parameters():
            p.grad = None
        s = m.score_packed(x, LENS)
        (s * w).sum().backward()
        return [s.detach().clone()] + [p.grad.clone() for p in m.parameters()]
    _same(*_both(monkeypatch, run), f"VASNet gradients {precision}")


@pytest.mark.parametrize("kind", ["dsn", "slstm"])
def test_lstm_scorers(dev, monkeypatch, kind):
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.models.sumgan import sLSTM
    torch.manual_seed(3)
    D = 128
    x = _x(D, dev, 2)
    m = (DSN(D, 256, 1) if kind == "dsn" else sLSTM(D, 320, 2)).to(dev).eval()       # H = 256: persistent kernels; 320: the wide ones
    w = torch.rand(sum(LENS), device=dev)

    def score():
        with torch.no_grad():
            return m.score_packed(x, LENS).clone()
    _same(*_both(monkeypatch, score), f"{kind} scoring")

    def grads():
        for p in m.parameters():
            p.grad = None
        s = m.score_packed(x, LENS)
        (s * w).sum().backward()
        return [s.detach().clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
    _same(*_both(monkeypatch, grads), f"{kind} gradients")


@pytest.mark.parametrize("precision", ["fp32", "bf16x6", "bf16x3"])
def test_transformer(dev, monkeypatch, precision):
    from summarizer_amd.models.transformer import Transformer
    torch.manual_seed(4)
    D = 256
    x = _x(D, dev, 3)
    xl = _x(D, dev, 5, LENS_LONG)
    for heads in (2, 4):                        # heads of 128 columns: attention on planes in the split modes; 64: in-loop products
        m = Transformer(input_size=D, encoder_layers=2, attention_heads=heads).to(dev).eval()
        m.precision = precision
        for xx, lens in ((x, LENS), (xl, LENS_LONG)):
            def score():
                m._wpl = None
                with torch.no_grad():
                    return m.score_packed(xx, lens).clone()
            _same(*_both(monkeypatch, score), f"Transformer {precision} heads={heads} lens={lens}")
    if precision == "fp32":
        w = torch.rand(sum(LENS), device=dev)

        def grads():
            for p in m.parameters():
                p.grad = None
            s = m.score_packed(x, LENS)
            (s * w).sum().backward()
            return [s.detach().clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
        _same(*_both(monkeypatch, grads), "Transformer gradients")


def test_dsn_reward(dev, monkeypatch):
    from summarizer_amd import kernels
    torch.manual_seed(5)
    D = 128
    x = _x(D, dev, 4)
    sb = kernels.SeqBatch.get(LENS, dev)
    actions = (torch.rand(5, sum(LENS), device=dev) < 0.4).float()

    def reward():
        return kernels.dsn_reward(x, sb, actions).clone()
    _same(*_both(monkeypatch, reward), "DSN reward")


@pytest.mark.parametrize("precision", ["fp32", "bf16x6", "bf16x3", "bf16"])
def test_vasnet_at_baseline_size(dev, monkeypatch, precision):
    """The S-TVSum batch (50 videos, 12 003 frames, D = 1024) takes other kernel instances than the small batches above (128 x 128 lean
    tiles, XCD maps, the wide bf16 kernels): scoring in every arithmetic and the training gradients in fp32 / bf16."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from summarizer_amd import kernels
    from summarizer_amd.models.vasnet import VASNet
    lens = bench.tvsum_lens(50)
    torch.manual_seed(6)
    x = _x(1024, dev, 6, lens)
    m = VASNet(input_size=1024, precision=precision).to(dev).eval()
    if precision != "bf16":
        def score():
            kernels.drop_shadows(x)
            m._wpl = None
            with torch.no_grad():
                return m.score_packed(x, lens).clone()
        _same(*_both(monkeypatch, score), f"VASNet S-TVSum {precision}")
    if precision in ("fp32", "bf16"):
        w = torch.rand(sum(lens), device=dev)

        def grads():
            kernels.drop_shadows(x)
            for p in m.parameters():
                p.grad = None
            s = m.score_packed(x, lens)
            (s * w).sum().backward()
            return [s.detach().clone()] + [p.grad.clone() for p in m.parameters()]
        _same(*_both(monkeypatch, grads), f"VASNet S-TVSum gradients {precision}")
