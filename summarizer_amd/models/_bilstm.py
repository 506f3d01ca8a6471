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
    precision = getattr(model, "precision", "fp32")     # "fp32" | "bf16x3" | "bf16x6" (kernels.precision_code)
    training = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
    if training:
        from ..autograd import BiLstmScorerFunction
        names = [n for n, _ in model.named_parameters()]
        p = dict(model.named_parameters())
        return BiLstmScorerFunction.apply(xp, sb, prefix, num_layers, H, head_w, head_b, precision,
                                          getattr(model, "tail_grads_ready_event", None), names, *[p[n] for n in names])
    p = dict(model.named_parameters())
    h = xp
    wpl = _layer_wplanes(model, p, prefix, num_layers, xp.shape[1], H, precision) if sb.n_rows >= 1024 else None
    for layer in range(num_layers):
        h, _ = kernels.bilstm_layer_forward(h, sb, p, prefix, layer, H, training=False, precision=precision,
                                            wplanes=None if wpl is None else wpl[layer], dataset_input=(layer == 0))
    return kernels.frame_head_forward(h, p[head_w], p[head_b])


def _layer_wplanes(model, p, prefix, num_layers, In, H, precision):
    """Per-layer weight-plane blocks of the input projections for the split-bf16 arithmetics (kernels.bilstm_wplanes), cached on the
    model and rebuilt when a weight changes (storage addresses, tensor versions, kernels.WEIGHTS_EPOCH: the rules of VASNet._wplanes)."""
    n_planes = kernels.PLANES_OF.get(precision)
    if not n_planes:
        return None
    ps = [v for k, v in sorted(p.items()) if k.startswith(prefix)]
    key = tuple(t.data_ptr() for t in ps) + tuple(t._version for t in ps) + (kernels.WEIGHTS_EPOCH[0], precision,
                                                                               torch.cuda.current_stream(ps[0].device).cuda_stream if ps[0].is_cuda else 0)
    cache = model.__dict__.get("_sumk_wpl")
    if cache is None or cache[0] != key:
        with torch.no_grad():
            old = cache[2] if cache is not None and cache[0][-1] == key[-1] else [None] * num_layers      # (VASNet._wplanes: no re-use across streams)
            blocks = [kernels.bilstm_wplanes({k: v.detach() for k, v in p.items()}, prefix, layer, In if layer == 0 else 2 * H, H, n_planes,
                                             out=old[layer]) for layer in range(num_layers)]
        cache = (key, blocks, [getattr(b, "_sumk_keep", None) if b is not None else None for b in blocks])
        model.__dict__["_sumk_wpl"] = cache
    return cache[1]


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


# ------------------------------------------------------------------------------------------------ GRU (DSN(cell="gru"))
def _time_major_index(sb, reverse):
    """(idx (T_max, B) int64 rows of the packed batch visited at step t -- the dummy row n_rows where the video has ended --,
    mask (T_max, B) float32).  Forward direction visits row off + t, reverse off + T - 1 - t.  Cached on the SeqBatch."""
    key = "_tm_rev" if reverse else "_tm_fwd"
    hit = getattr(sb, key, None)
    if hit is None:
        import numpy as np
        lens = np.asarray(sb.lens); off = sb.off_host[:-1].astype(np.int64)
        t = np.arange(lens.max())[:, None]
        active = t < lens[None, :]
        rows = off[None, :] + (lens[None, :] - 1 - t if reverse else t)
        idx = np.where(active, rows, sb.n_rows)
        hit = (torch.from_numpy(idx).to(sb.device), torch.from_numpy(active.astype(np.float32)).to(sb.device))
        setattr(sb, key, hit)
    return hit


class GruLayerFunction(torch.autograd.Function):
    """One bidirectional nn.GRU layer on a packed batch: x (n_rows, In) -> h (n_rows, 2H) = [h_fwd | h_rev].
    params: (w_ih, w_hh, b_ih, b_hh) of the forward direction, then of the reverse one (nn.GRU's names / shapes).
    The host walks the steps (time-major, ended videos masked); per step the recurrent projection is an MFMA GEMM
    (`sumk_linear_forward`) and the gates one fused kernel (`sumk_gru_cell_forward`); BPTT mirrors it step by step and the
    input-side gradients are three GEMM calls at the end.  Functional, not tuned (csrc/gru.hip)."""

    @staticmethod
    def forward(ctx, xp, sb, H, precision, *params):
        R = sb.n_rows
        outs, saved = [], []
        for d in range(2):
            w_ih, w_hh, b_ih, b_hh = params[4 * d:4 * d + 4]
            idx, mask = _time_major_index(sb, reverse=(d == 1))
            gx_all = kernels.linear_forward(xp, w_ih, b_ih, precision)                    # (R, 3H)
            gx_pad = torch.cat([gx_all, gx_all.new_zeros(1, 3 * H)])                       # + dummy row for ended videos
            out = xp.new_zeros(R + 1, H)
            h = xp.new_zeros(sb.n_seq, H)
            steps = []
            for t in range(idx.shape[0]):
                gx_t = gx_pad.index_select(0, idx[t])
                gh_t = kernels.linear_forward(h, w_hh, b_hh, precision)
                h_new, rzn = kernels.gru_cell_forward(gx_t, gh_t, h, mask[t], save=True)
                out.index_copy_(0, idx[t], h_new)
                steps.append((h, gh_t, rzn))
                h = h_new
            outs.append(out[:R])
            saved.append(steps)
        ctx.meta = (sb, H, precision)
        ctx.steps, ctx.params = saved, params
        ctx.save_for_backward(xp)
        return torch.cat(outs, dim=1)

    @staticmethod
    def backward(ctx, dout):
        from ..autograd import _grad_targets
        (xp,) = ctx.saved_tensors
        sb, H, precision = ctx.meta
        params = ctx.params
        grads, ret = _grad_targets([str(i) for i in range(8)], params)
        R = sb.n_rows
        dx = None
        for d in range(2):
            w_ih, w_hh = params[4 * d], params[4 * d + 1]
            g_wih, g_whh, g_bih, g_bhh = (grads[str(4 * d + i)] for i in range(4))
            idx, mask = _time_major_index(sb, reverse=(d == 1))
            do = torch.cat([dout[:, d * H:(d + 1) * H].contiguous(), dout.new_zeros(1, H)])     # dummy row: zero gradient
            dgx_all = xp.new_zeros(R + 1, 3 * H)
            dh_carry = xp.new_zeros(sb.n_seq, H)
            for t in range(idx.shape[0] - 1, -1, -1):
                h_prev, gh_t, rzn = ctx.steps[d][t]
                dh = do.index_select(0, idx[t]) + dh_carry
                dgx_t, dgh_t, dh_direct = kernels.gru_cell_backward(dh, rzn, gh_t, h_prev, mask[t])
                dgx_all.index_copy_(0, idx[t], dgx_t)
                dh_rec = kernels.linear_backward(h_prev, w_hh, dgh_t, g_whh, g_bhh, True, precision)   # dW_hh, db_hh accumulate
                dh_carry = dh_direct + dh_rec
            part = kernels.linear_backward(xp, w_ih, dgx_all[:R].contiguous(), g_wih, g_bih, ctx.needs_input_grad[0], precision)
            if part is not None:
                dx = part if dx is None else dx + part
        ctx.steps = ctx.params = None
        return (dx, None, None, None) + tuple(ret)


def bigru_scores(model, xp, sb, num_layers, H, head_w, head_b):
    """Stacked bidirectional GRU + per-frame Linear(2H, 1) + sigmoid (DSN(cell="gru"), dsn.py:28-47)."""
    from ..autograd import FrameHeadFunction
    precision = getattr(model, "precision", "fp32")
    p = dict(model.named_parameters())
    h = xp
    for l in range(num_layers):
        names = [f"rnn.{n}_l{l}{suf}" for suf in ("", "_reverse") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        h = GruLayerFunction.apply(h, sb, H, precision, *[p[n] for n in names])
    return FrameHeadFunction.apply(h, p[head_w], p[head_b])
