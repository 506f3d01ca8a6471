"""numpy restatement of VASNet.forward (reference: summarizer/models/vasnet.py:92-148).  TEST INFRASTRUCTURE ONLY.

Parameters are passed as a dict keyed exactly like the reference state_dict
(vasnet.py:56-66 / SURVEY section 5): K.weight, Q.weight, V.weight, attention_head_projection.weight,
k1.weight, k1.bias, k2.weight, k2.bias, layer_norm.weight, layer_norm.bias [, pos_embed.weight].
"""
import numpy as np


def layer_norm(y, w, b, eps):
    # torch.nn.LayerNorm over the last dim, biased variance (vasnet.py:54,137,143)
    mu = y.mean(axis=-1, keepdims=True)
    var = ((y - mu) ** 2).mean(axis=-1, keepdims=True)
    return (y - mu) / np.sqrt(var + eps) * w + b


def sinusoid_table(max_length, d):
    # vasnet.py:44-48 -- note the reference's exponent is 2*i/d for even i and 2*(i+1)/d for odd (i+1)
    tab = np.zeros((max_length, d), dtype=np.float32)
    pos = np.arange(max_length, dtype=np.float64)[:, None]
    i = np.arange(0, d, 2, dtype=np.float64)[None, :]
    tab[:, 0::2] = np.sin(pos / (10000.0 ** ((2 * i) / d)))
    tab[:, 1::2] = np.cos(pos / (10000.0 ** ((2 * (i + 1)) / d)))
    return tab


def pos_rows(T, B, kind):
    """Row of the position table added to frame t of batch element b -> (B,T) int array.

    "simple"    (vasnet.py:108-109): arange(T).repeat(1,B).view(B,T)            -> t
    "attention" (vasnet.py:111): table[:T].repeat(1,B).view(B,T,D) re-views a (T, B*D) buffer as (B,T,D),
                 so element (b,t) receives table row (b*T + t) // B -- equal to t only when B == 1.
                 (A reference quirk; reproduced, not fixed.)
    """
    b = np.arange(B)[:, None]; t = np.arange(T)[None, :]
    return np.broadcast_to(t, (B, T)).copy() if kind == "simple" else (b * T + t) // B


def attention_mask(e, ignore_self, aperture):
    """In-place masking of one (T,T) logits matrix, exactly in the reference's order (vasnet.py:121-127)."""
    T = e.shape[0]
    if ignore_self:
        e[np.eye(T, dtype=bool)] = -np.inf
    if aperture is not None:
        with np.errstate(invalid="ignore", over="ignore", under="ignore"):
            scope = np.tril(e, aperture) * np.triu(e, -aperture)
        e[scope == 0] = -np.inf
    return e


def vasnet_forward(x, p, ignore_self=False, aperture=None, scale=None, eps=1e-6,
                   pos_table=None, pos_kind="simple", dtype=np.float32, return_intermediates=False):
    """x: (T, B, D) -> (T, B, 1).  Eval mode (dropout = identity, vasnet.py:130,136,142).

    pos_table: (max_length, D) table added to x (vasnet.py:106-112); the reference adds IN PLACE on a
    view of the caller's tensor, so the residual (vasnet.py:135) sees x + pos.  Here x is not mutated.
    """
    x = np.asarray(x)
    T, B, D = x.shape
    f = dtype
    xb = np.ascontiguousarray(np.transpose(x, (1, 0, 2))).astype(f)  # (B,T,D) vasnet.py:99
    if pos_table is not None:
        assert pos_table.shape[0] >= T, "input sequence has higher length than max_length"
        xb = xb + pos_table.astype(f)[pos_rows(T, B, pos_kind)]
    sc = f(scale if scale is not None else 1.0 / np.sqrt(D))   # vasnet.py:34
    W = {k: np.asarray(v).astype(f) for k, v in p.items()}
    K = xb @ W["K.weight"].T
    Q = xb @ W["Q.weight"].T
    V = xb @ W["V.weight"].T
    out = np.empty((B, T, 1), dtype=f)
    inter = {}
    for b in range(B):
        e = (Q[b] @ K[b].T) * sc                              # vasnet.py:118-119
        e = attention_mask(e, ignore_self, aperture)
        with np.errstate(invalid="ignore"):
            m = e.max(axis=1, keepdims=True)
            a = np.exp(e - m)
            alpha = a / a.sum(axis=1, keepdims=True)          # softmax over keys, vasnet.py:129
        c = alpha @ V[b]                                       # vasnet.py:131
        c = c @ W["attention_head_projection.weight"].T        # vasnet.py:132
        y = c + xb[b]                                          # vasnet.py:135
        y1 = layer_norm(y, W["layer_norm.weight"], W["layer_norm.bias"], f(eps))   # vasnet.py:137
        z = y1 @ W["k1.weight"].T + W["k1.bias"]               # vasnet.py:140
        z = np.maximum(z, 0)                                   # vasnet.py:141
        y2 = layer_norm(z, W["layer_norm.weight"], W["layer_norm.bias"], f(eps))   # SAME LN, vasnet.py:143
        s = y2 @ W["k2.weight"].T + W["k2.bias"]               # vasnet.py:144
        out[b] = 1.0 / (1.0 + np.exp(-s))                      # vasnet.py:145
        if return_intermediates and b == 0:
            inter = dict(Q=Q[b], K=K[b], V=V[b], e=e, alpha=alpha, c=c, y1=y1, z=z, y2=y2)
    y = np.transpose(out, (1, 0, 2))                           # vasnet.py:147
    return (y, inter) if return_intermediates else y
