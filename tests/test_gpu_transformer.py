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
    from summarizer_amd._lib import SumkError
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
        with pytest.raises(SumkError):
            m(x)                                   # grad enabled: inference-only scorer refuses loudly
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
