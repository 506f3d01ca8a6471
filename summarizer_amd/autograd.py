"""torch.autograd glue: lets `loss.backward()` in a reference-style training loop (vasnet.py:207-212,
dsn.py:111-146) reach the HIP backward kernels.  Only plumbing lives here: every gradient is produced by
libsumk.so; torch supplies the tensors that hold them."""
import torch

from . import kernels


def _grad_targets(names, params):
    """Gradient buffers for the HIP backward (which ACCUMULATES).  A parameter that already owns a contiguous .grad
    (FlatAdam's bucket views) is accumulated into directly -- no zeros_like + add kernels, no extra 21 MB of traffic --
    and autograd is handed None for it; otherwise a fresh zero buffer is returned to autograd as usual."""
    grads, ret = {}, []
    for n, p in zip(names, params):
        g = getattr(p, "grad", None)
        if g is not None and g.is_contiguous() and g.dtype == p.dtype and g.shape == p.shape:
            grads[n] = g; ret.append(None)
        else:
            grads[n] = torch.zeros_like(p); ret.append(grads[n])
    return grads, ret


class VasnetFunction(torch.autograd.Function):
    """scores = VASNet(x) for a packed batch.  inputs: x, SeqBatch, opts, pos table/rows, param names, *params."""

    @staticmethod
    def forward(ctx, xp, sb, opts, table, rows, names, *params):
        p = dict(zip(names, params))
        scores, ws = kernels.vasnet_forward_packed(xp, sb, p, opts, table, rows, training=True)
        ctx.sb, ctx.opts, ctx.names, ctx.ws, ctx.rows = sb, opts, names, ws, rows
        ctx.table_is_param = isinstance(table, torch.nn.Parameter) and table.requires_grad
        ctx.table = table
        ctx.params = params          # the Parameter objects themselves (their .grad may be a bucket view)
        ctx.save_for_backward(xp)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        (xp,) = ctx.saved_tensors
        params = ctx.params
        p = dict(zip(ctx.names, params))
        grads, ret = _grad_targets(ctx.names, params)
        want_dx = ctx.needs_input_grad[0] or ctx.table_is_param
        dx = kernels.vasnet_backward_packed(xp, ctx.sb, p, ctx.opts, dscores, ctx.ws, grads, want_dx=want_dx)
        ctx.ws = None
        if ctx.table_is_param:
            # learnable positional embedding (vasnet.py:42): rows of dx scatter-add into the table.  A few KB of
            # index bookkeeping, done with a torch op rather than a dedicated kernel.
            tg = torch.zeros_like(ctx.table).index_add_(0, ctx.rows.long(), dx)
            ctx.table.grad = tg if ctx.table.grad is None else ctx.table.grad + tg
        gx = dx if ctx.needs_input_grad[0] else None
        ctx.params = None
        return (gx, None, None, None, None, None) + tuple(ret)


class BiLstmScorerFunction(torch.autograd.Function):
    """scores = sigmoid(Linear(BiLSTM_stack(x))) for a packed batch (DSN: dsn.py:45-46, sLSTM: sumgan.py:43-45)."""

    @staticmethod
    def forward(ctx, xp, sb, prefix, num_layers, H, head_w, head_b, precision, names, *params):
        p = dict(zip(names, params))
        acts, wss = [xp], []
        for layer in range(num_layers):
            h, ws = kernels.bilstm_layer_forward(acts[-1], sb, p, prefix, layer, H, training=True, precision=precision)
            acts.append(h); wss.append(ws)
        scores = kernels.frame_head_forward(acts[-1], p[head_w], p[head_b])
        ctx.meta = (sb, prefix, num_layers, H, head_w, head_b, precision, names)
        ctx.wss = wss
        ctx.params = params
        ctx.save_for_backward(scores, *acts)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        sb, prefix, num_layers, H, head_w, head_b, precision, names = ctx.meta
        saved = ctx.saved_tensors
        scores, acts, params = saved[0], saved[1:2 + num_layers], ctx.params
        p = dict(zip(names, params))
        grads, ret = _grad_targets(names, params)
        dh = kernels.frame_head_backward(acts[-1], scores, dscores, p[head_w], grads[head_w], grads[head_b])
        for layer in range(num_layers - 1, -1, -1):
            want_dx = layer > 0 or ctx.needs_input_grad[0]
            dh = kernels.bilstm_layer_backward(acts[layer], acts[layer + 1], dh, sb, p, grads, prefix, layer, H,
                                               ctx.wss[layer], want_dx, precision=precision)
        ctx.wss = None
        gx = dh if ctx.needs_input_grad[0] else None
        ctx.params = None
        return (gx, None, None, None, None, None, None, None, None) + tuple(ret)


class TransformerFunction(torch.autograd.Function):
    """scores = Transformer-encoder scorer(x) for a packed batch (transformer.py:74-103)."""

    @staticmethod
    def forward(ctx, xp, sb, cfg, opts, table, rows, names, *params):
        p = dict(zip(names, params))
        scores, ws = kernels.transformer_forward_packed(xp, sb, p, cfg["n_layers"], cfg["n_heads"], cfg["dff"], opts, table,
                                                        rows, training=True)
        ctx.meta = (sb, cfg, opts, names, rows)
        ctx.ws, ctx.params, ctx.table = ws, params, table
        ctx.table_is_param = isinstance(table, torch.nn.Parameter) and table.requires_grad
        ctx.save_for_backward(xp)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        (xp,) = ctx.saved_tensors
        sb, cfg, opts, names, rows = ctx.meta
        p = dict(zip(names, ctx.params))
        grads, ret = _grad_targets(names, ctx.params)
        want_dx = ctx.needs_input_grad[0] or ctx.table_is_param
        dx = kernels.transformer_backward_packed(xp, sb, p, grads, cfg["n_layers"], cfg["n_heads"], cfg["dff"], opts, dscores,
                                                 ctx.ws, want_dx=want_dx)
        if ctx.table_is_param:
            tg = torch.zeros_like(ctx.table).index_add_(0, rows.long(), dx)
            ctx.table.grad = tg if ctx.table.grad is None else ctx.table.grad + tg
        ctx.ws = ctx.params = None
        return (dx if ctx.needs_input_grad[0] else None, None, None, None, None, None, None) + tuple(ret)
