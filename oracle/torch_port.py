"""Stock-PyTorch (CPU, fp32) functional port of the reference scorers.  TEST INFRASTRUCTURE ONLY.

This is (1) a second, autograd-capable checker for the HIP path (forward AND gradients), and
(2) the `cpu_baseline` ("kind": "port") that bench.py times on the GPU node's host cores: it issues the
same ATen op sequence as the reference modules (Linear / bmm / softmax / layer_norm / nn.LSTM), one video
per call, exactly like Trainer.test (summarizer/models/__init__.py:45-54).
Pinned against the real reference by tests/golden/*.npz (see tests/test_oracle.py).
"""
import math
import torch
import torch.nn.functional as F


def _r16(t):
    """fp32 -> bf16 (round to nearest even, what v_cvt_pk_bf16_f32 does) -> fp32"""
    return t.to(torch.bfloat16).to(torch.float32)


class _Linear16(torch.autograd.Function):
    """y = x W^T with BOTH operands rounded to bf16 and fp32 accumulation -- and the same in the backward: dx = bf16(dy) bf16(W),
    dW = bf16(dy)^T bf16(x).  This is the arithmetic of the mixed-precision training step (csrc/gemm_b16.hip: every GEMM reads bf16
    shadows of its operands, gradients included), restated with stock fp32 matmuls (a product of two bf16 values is exact in fp32)."""

    @staticmethod
    def forward(ctx, x, w):
        xr, wr = _r16(x), _r16(w)
        ctx.save_for_backward(xr, wr)
        return xr @ wr.t()

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = _r16(dy)
        return dyr @ wr, dyr.reshape(-1, dyr.shape[-1]).t() @ xr.reshape(-1, xr.shape[-1])


class _Bmm16(torch.autograd.Function):
    """C = A B per batch entry, operands rounded to bf16 in the forward and in both backward products."""

    @staticmethod
    def forward(ctx, a, b):
        ar, br = _r16(a), _r16(b)
        ctx.save_for_backward(ar, br)
        return torch.bmm(ar, br)

    @staticmethod
    def backward(ctx, dy):
        ar, br = ctx.saved_tensors
        dyr = _r16(dy)
        return torch.bmm(dyr, br.transpose(1, 2)), torch.bmm(ar.transpose(1, 2), dyr)


def vasnet_scores(x, p, ignore_self=False, aperture=None, scale=None, eps=1e-6, pos_table=None,
                  pos_kind="simple", drop_masks=None, return_logits=False, bf16_products=False):
    """x: (T,B,D) -> (T,B,1).  Op order of vasnet.py:99-147.

    drop_masks: optional (m_alpha (B,T,T), m_y (B,T,D), m_z (B,T,D)) of already-scaled keep masks
    (0 or 1/(1-p)) so training-mode dropout can be checked deterministically.
    return_logits: also return the pre-sigmoid outputs of k2 (vasnet.py:144).
    bf16_products: emulate the mixed-precision training arithmetic (see _Linear16) instead of fp32 products.
    """
    T, B, D = x.shape
    xb = x.permute(1, 0, 2)
    if pos_table is not None:
        bi = torch.arange(B).unsqueeze(1); ti = torch.arange(T).unsqueeze(0)
        rows = ti.expand(B, T) if pos_kind == "simple" else (bi * T + ti) // B     # vasnet.py:108-111 (see vasnet_np.pos_rows)
        xb = xb + pos_table[rows]
    sc = scale if scale is not None else 1.0 / math.sqrt(D)
    # bf16_products: the emulated mixed-precision arithmetic (precision="bf16"): every matrix product on bf16-rounded operands with
    # fp32 accumulation, forward and backward; softmax / LayerNorm / residual / head stay fp32 as in csrc/vasnet.hip
    lin = (lambda a, w_, b_=None: _Linear16.apply(a, w_) + (b_ if b_ is not None else 0.0)) if bf16_products else F.linear
    bmm = _Bmm16.apply if bf16_products else torch.bmm
    K = lin(xb, p["K.weight"]); Q = lin(xb, p["Q.weight"]); V = lin(xb, p["V.weight"])
    e = bmm(Q, K.transpose(1, 2)) * sc
    if ignore_self:
        e = e.masked_fill(torch.eye(T, dtype=torch.bool, device=e.device).unsqueeze(0), float("-inf"))
    if aperture is not None:
        scope = torch.tril(e, diagonal=aperture) * torch.triu(e, diagonal=-aperture)
        e = e.masked_fill(scope == 0, float("-inf"))
    alpha = torch.softmax(e, dim=2)
    if drop_masks is not None:
        alpha = alpha * drop_masks[0]
    c = lin(bmm(alpha, V), p["attention_head_projection.weight"])
    y = c + xb
    if drop_masks is not None:
        y = y * drop_masks[1]
    y = F.layer_norm(y, (D,), p["layer_norm.weight"], p["layer_norm.bias"], eps)
    z = torch.relu(lin(y, p["k1.weight"], p["k1.bias"]))
    if drop_masks is not None:
        z = z * drop_masks[2]
    z = F.layer_norm(z, (D,), p["layer_norm.weight"], p["layer_norm.bias"], eps)
    u = F.linear(z, p["k2.weight"], p["k2.bias"])
    s = torch.sigmoid(u)
    if return_logits:            # (scores, pre-sigmoid k2 outputs): where scores saturate, the logits still tell two paths apart
        return s.permute(1, 0, 2), u.permute(1, 0, 2)
    return s.permute(1, 0, 2)


def make_lstm(p, prefix, input_size, hidden_size, num_layers):
    """Builds a stock nn.LSTM carrying the given weights (dsn.py:23-27 / sumgan.py:27-32)."""
    m = torch.nn.LSTM(input_size, hidden_size, num_layers=num_layers, bidirectional=True)
    sd = {k[len(prefix):]: torch.as_tensor(v) for k, v in p.items() if k.startswith(prefix)}
    m.load_state_dict(sd)
    return m


def bilstm_scores(x, p, prefix, head_w, head_b, input_size, hidden_size, num_layers, lstm=None):
    lstm = lstm if lstm is not None else make_lstm(p, prefix, input_size, hidden_size, num_layers)
    h, _ = lstm(x)
    return torch.sigmoid(F.linear(h, p[head_w], p[head_b]))


class TransformerPort(torch.nn.Module):
    """Stock-PyTorch restatement of the reference Transformer scorer (transformer.py:19-103) with every dropout at 0:
    learnable positional table added in place, nn.TransformerEncoder (post-norm, FF = D) whose final norm is the SAME
    LayerNorm that follows k1, then k1 / ReLU / LayerNorm / k2 / sigmoid.  Autograd-capable checker for the HIP path."""

    def __init__(self, D, n_layers, n_heads, max_length=None, eps=1e-5, more_residuals=False):
        super().__init__()
        nn = torch.nn
        self.pos_embed = nn.Embedding(max_length, D) if max_length else None
        self.layer_norm = nn.LayerNorm(D, eps)
        layer = nn.TransformerEncoderLayer(d_model=D, nhead=n_heads, dim_feedforward=D, dropout=0.0, activation="relu")
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=n_layers, norm=self.layer_norm, enable_nested_tensor=False)
        self.k1, self.k2 = nn.Linear(D, D), nn.Linear(D, 1)
        self.more_residuals = more_residuals

    def forward(self, x):
        T, B, D = x.shape
        if self.pos_embed is not None:
            x = x + self.pos_embed(torch.arange(T)).unsqueeze(1)          # 'simple' table: row t for every batch entry
        e = self.transformer_encoder(x)
        if self.more_residuals:
            e = e + x
        return torch.sigmoid(self.k2(self.layer_norm(torch.relu(self.k1(e)))))


def make_gru(p, prefix, input_size, hidden_size, num_layers):
    """Stock nn.GRU carrying the given weights (the reference's optional DSN(cell="gru"), dsn.py:28-33)."""
    m = torch.nn.GRU(input_size, hidden_size, num_layers=num_layers, bidirectional=True)
    m.load_state_dict({k[len(prefix):]: torch.as_tensor(v) for k, v in p.items() if k.startswith(prefix)})
    return m


def bigru_scores(x, p, prefix, head_w, head_b, gru):
    h, _ = gru(x)
    return torch.sigmoid(F.linear(h, p[head_w], p[head_b]))
