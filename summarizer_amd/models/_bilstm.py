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


def lstm_stack(lstm, xp, sb, h0=None, c0=None, precision="fp32"):
    """nn.LSTM(bidirectional=False) container `lstm` applied to packed rows xp: -> (out (n_rows, H), h_n, c_n (L, n_seq, H)).
    The nn.LSTM object only holds the parameters (same names / shapes / init as the reference); it is never called."""
    H, L = lstm.hidden_size, lstm.num_layers
    params = []
    for l in range(L):
        params += [getattr(lstm, f"weight_ih_l{l}"), getattr(lstm, f"weight_hh_l{l}"), getattr(lstm, f"bias_ih_l{l}"),
                   getattr(lstm, f"bias_hh_l{l}")]
    needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or xp.requires_grad or
                                              (h0 is not None and h0.requires_grad) or (c0 is not None and c0.requires_grad))
    if needs_grad:
        from ..autograd import LstmStackFunction
        return LstmStackFunction.apply(xp, sb, H, precision, h0, c0, *params)
    h, hn, cn = xp, [], []
    for l in range(L):
        h, hl, cl, _ = kernels.lstm_layer_forward(h, sb, params[4 * l:4 * l + 4], H, None if h0 is None else h0[l].contiguous(),
                                                  None if c0 is None else c0[l].contiguous(), training=False, precision=precision)
        hn.append(hl); cn.append(cl)
    return h, torch.stack(hn), torch.stack(cn)
