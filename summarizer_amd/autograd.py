"""torch.autograd glue: lets `loss.backward()` in a reference-style training loop (vasnet.py:207-212,
dsn.py:111-146) reach the HIP backward kernels.  Only plumbing lives here: every gradient is produced by
libsumk.so; torch supplies the tensors that hold them."""
import torch

from . import kernels


class VasnetFunction(torch.autograd.Function):
    """scores = VASNet(x) for a packed batch.  inputs: x, SeqBatch, opts, pos table/rows, param names, *params."""

    @staticmethod
    def forward(ctx, xp, sb, opts, table, rows, names, *params):
        p = dict(zip(names, params))
        scores, ws = kernels.vasnet_forward_packed(xp, sb, p, opts, table, rows, training=True)
        ctx.sb, ctx.opts, ctx.names, ctx.ws, ctx.rows = sb, opts, names, ws, rows
        ctx.table_is_param = isinstance(table, torch.nn.Parameter) and table.requires_grad
        ctx.table = table
        ctx.save_for_backward(xp, *params)
        ctx.mark_non_differentiable()
        return scores

    @staticmethod
    def backward(ctx, dscores):
        xp, *params = ctx.saved_tensors
        p = dict(zip(ctx.names, params))
        grads = {k: torch.zeros_like(v) for k, v in p.items()}
        want_dx = ctx.needs_input_grad[0] or ctx.table_is_param
        dx = kernels.vasnet_backward_packed(xp, ctx.sb, p, ctx.opts, dscores, ctx.ws, grads, want_dx=want_dx)
        ctx.ws = None
        if ctx.table_is_param:
            # learnable positional embedding (vasnet.py:42): rows of dx scatter-add into the table.  A few KB of
            # index bookkeeping, done with a torch op rather than a dedicated kernel.
            tg = torch.zeros_like(ctx.table).index_add_(0, ctx.rows.long(), dx)
            ctx.table.grad = tg if ctx.table.grad is None else ctx.table.grad + tg
        gx = dx if ctx.needs_input_grad[0] else None
        return (gx, None, None, None, None, None) + tuple(grads[n] for n in ctx.names)


class BiLstmScorerFunction(torch.autograd.Function):
    """scores = sigmoid(Linear(BiLSTM_stack(x))) for a packed batch (DSN: dsn.py:45-46, sLSTM: sumgan.py:43-45)."""

    @staticmethod
    def forward(ctx, xp, sb, prefix, num_layers, H, head_w, head_b, names, *params):
        p = dict(zip(names, params))
        acts, wss = [xp], []
        for layer in range(num_layers):
            h, ws = kernels.bilstm_layer_forward(acts[-1], sb, p, prefix, layer, H, training=True)
            acts.append(h); wss.append(ws)
        scores = kernels.frame_head_forward(acts[-1], p[head_w], p[head_b])
        ctx.meta = (sb, prefix, num_layers, H, head_w, head_b, names)
        ctx.wss = wss
        ctx.save_for_backward(scores, *acts, *params)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        sb, prefix, num_layers, H, head_w, head_b, names = ctx.meta
        saved = ctx.saved_tensors
        scores, acts, params = saved[0], saved[1:2 + num_layers], saved[2 + num_layers:]
        p = dict(zip(names, params))
        grads = {k: torch.zeros_like(v) for k, v in p.items()}
        dh = kernels.frame_head_backward(acts[-1], scores, dscores, p[head_w], grads[head_w], grads[head_b])
        for layer in range(num_layers - 1, -1, -1):
            want_dx = layer > 0 or ctx.needs_input_grad[0]
            dh = kernels.bilstm_layer_backward(acts[layer], acts[layer + 1], dh, sb, p, grads, prefix, layer, H,
                                               ctx.wss[layer], want_dx)
        ctx.wss = None
        gx = dh if ctx.needs_input_grad[0] else None
        return (gx, None, None, None, None, None, None, None) + tuple(grads[n] for n in names)
