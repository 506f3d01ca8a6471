"""numpy restatement of the reference's Transformer-encoder scorer (summarizer/models/transformer.py:74-103), whose
encoder is stock `nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=D, nhead, dim_feedforward=D, relu))`
(transformer.py:49-50: post-norm layers, final norm = the SHARED `layer_norm`, which is applied again after k1).
TEST INFRASTRUCTURE ONLY.  Eval mode (all dropouts off).  Weights keyed like the reference state_dict."""
import numpy as np

from .vasnet_np import layer_norm, pos_rows


def _mha(x, p, pre, n_heads, f):
    """torch.nn.MultiheadAttention self-attention on one sequence x (T, D)."""
    T, D = x.shape
    dh = D // n_heads
    qkv = x @ p[pre + "self_attn.in_proj_weight"].astype(f).T + p[pre + "self_attn.in_proj_bias"].astype(f)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    q = q * f(1.0 / np.sqrt(dh))                          # torch scales q before the product
    out = np.empty((T, D), dtype=f)
    for h in range(n_heads):
        sl = slice(h * dh, (h + 1) * dh)
        e = q[:, sl] @ k[:, sl].T
        e = e - e.max(axis=1, keepdims=True)
        a = np.exp(e); a = a / a.sum(axis=1, keepdims=True)
        out[:, sl] = a @ v[:, sl]
    return out @ p[pre + "self_attn.out_proj.weight"].astype(f).T + p[pre + "self_attn.out_proj.bias"].astype(f)


def transformer_forward(x, p, n_layers, n_heads, eps=1e-5, more_residuals=False, pos_table=None, pos_kind="simple",
                        dtype=np.float32):
    """x: (T, B, D) -> (T, B, 1)."""
    f = dtype
    x = np.asarray(x)
    T, B, D = x.shape
    xb = np.ascontiguousarray(np.transpose(x, (1, 0, 2))).astype(f)
    if pos_table is not None:
        xb = xb + pos_table.astype(f)[pos_rows(T, B, pos_kind)]               # transformer.py:83-89 (same quirk as VASNet)
    g = lambda k: np.asarray(p[k]).astype(f)
    out = np.empty((B, T, 1), dtype=f)
    for b in range(B):
        h = xb[b]
        for l in range(n_layers):
            pre = f"transformer_encoder.layers.{l}."
            h = layer_norm(h + _mha(h, p, pre, n_heads, f), g(pre + "norm1.weight"), g(pre + "norm1.bias"), f(1e-5))
            ff = np.maximum(h @ g(pre + "linear1.weight").T + g(pre + "linear1.bias"), 0) @ g(pre + "linear2.weight").T \
                + g(pre + "linear2.bias")
            h = layer_norm(h + ff, g(pre + "norm2.weight"), g(pre + "norm2.bias"), f(1e-5))
        h = layer_norm(h, g("layer_norm.weight"), g("layer_norm.bias"), f(eps))   # encoder's final norm (shared LN)
        if more_residuals:
            h = h + xb[b]                                                        # transformer.py:94-95
        y = np.maximum(h @ g("k1.weight").T + g("k1.bias"), 0)                   # transformer.py:97-98
        y = layer_norm(y, g("layer_norm.weight"), g("layer_norm.bias"), f(eps))  # same LN again, transformer.py:100
        s = y @ g("k2.weight").T + g("k2.bias")
        out[b] = 1.0 / (1.0 + np.exp(-s))
    return np.transpose(out, (1, 0, 2))
