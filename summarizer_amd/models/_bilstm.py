"""Shared body of the two bidirectional-LSTM scorers (DSN, sLSTM): stacked BiLSTM layers + per-frame
Linear(2H,1)+Sigmoid head, all in libsumk.so.  The nn.LSTM / nn.Linear objects held by the model classes are
PARAMETER CONTAINERS only (identical names, shapes and default init as the reference); they are never called."""
import torch

from .. import kernels


def pack_time_major(x):
    """(T,B,F) time-major -> ((B*T,F) batch-major packed rows, lens).  Zero-copy for B == 1."""
    T, B, F = x.shape
    if B == 1 and x.is_contiguous():
        return x.view(T, F), [T]
    return x.permute(1, 0, 2).contiguous().view(B * T, F), [T] * B


def bilstm_scores(model, xp, sb, prefix, num_layers, H, head_w, head_b):
    precision = getattr(model, "precision", "fp32")     # "fp32" | "bf16x3" (kernels.precision_code)
    training = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
    if training:
        from ..autograd import BiLstmScorerFunction
        names = [n for n, _ in model.named_parameters()]
        p = dict(model.named_parameters())
        return BiLstmScorerFunction.apply(xp, sb, prefix, num_layers, H, head_w, head_b, precision, names, *[p[n] for n in names])
    p = dict(model.named_parameters())
    h = xp
    for layer in range(num_layers):
        h, _ = kernels.bilstm_layer_forward(h, sb, p, prefix, layer, H, training=False, precision=precision)
    return kernels.frame_head_forward(h, p[head_w], p[head_b])
