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
