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


def _table_grad(table, rows, dx):
    """Learnable positional embedding (vasnet.py:42, transformer.py:49): rows of dx scatter-add into the table's gradient.
    Accumulated IN PLACE when .grad exists -- FlatAdam's .grad is a view of its flat bucket, and rebinding .grad to a new
    tensor would leave the bucket (what step() and the all-reduce read) at zero.  A few KB of index bookkeeping, done
    with a torch op rather than a dedicated kernel."""
    with torch.no_grad():
        if table.grad is None:
            table.grad = torch.zeros_like(table)
        table.grad.index_add_(0, rows.long(), dx)


class VasnetFunction(torch.autograd.Function):
    """scores = VASNet(x) for a packed batch.  inputs: x, SeqBatch, opts, pos table/rows, param names, *params."""

    @staticmethod
    def forward(ctx, xp, sb, opts, table, rows, names, *params):
        p = dict(zip(names, params))
        opts = dict(opts)            # the call adds what it derives (problem tables, the bf16 shadow of x) for the backward pass; the caller's dict stays as it was
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
            _table_grad(ctx.table, ctx.rows, dx)
        gx = dx if ctx.needs_input_grad[0] else None
        ctx.params = None
        return (gx, None, None, None, None, None) + tuple(ret)


class BiLstmScorerFunction(torch.autograd.Function):
    """scores = sigmoid(Linear(BiLSTM_stack(x))) for a packed batch (DSN: dsn.py:45-46, sLSTM: sumgan.py:43-45)."""

    @staticmethod
    def forward(ctx, xp, sb, prefix, num_layers, H, head_w, head_b, precision, tail_event, names, *params):
        p = dict(zip(names, params))
        acts, wss = [xp], []
        ctx.tail_event = tail_event
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
            # (the event is passed for ONE-layer models only -- DSN: layer 0 is then the last layer processed and every other gradient of the
            #  bucket's tail is final; a deeper stack gets no early piece: its all-reduce simply starts after the backward)
            dh = kernels.bilstm_layer_backward(acts[layer], acts[layer + 1], dh, sb, p, grads, prefix, layer, H,
                                               ctx.wss[layer], want_dx, precision=precision,
                                               tail_event=ctx.tail_event if (layer == 0 and num_layers == 1) else None)
        ctx.wss = None
        gx = dh if ctx.needs_input_grad[0] else None
        ctx.params = None
        return (gx, None, None, None, None, None, None, None, None, None) + tuple(ret)


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
            _table_grad(ctx.table, rows, dx)
        ctx.ws = ctx.params = None
        return (dx if ctx.needs_input_grad[0] else None, None, None, None, None, None, None) + tuple(ret)


class LinearFunction(torch.autograd.Function):
    """y = x W^T + b on the MFMA GEMM (the small Linear layers around SumGAN's LSTM stacks, sumgan.py:58-59,84)."""

    @staticmethod
    def forward(ctx, x, w, b, precision):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        y = kernels.linear_forward(x2, w, b, precision)
        ctx.save_for_backward(x2, w)
        ctx.meta = (shape, b is not None, precision)
        ctx.params = (w, b)
        return y.view(*shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        shape, has_b, precision = ctx.meta
        names = ["w", "b"] if has_b else ["w"]
        grads, ret = _grad_targets(names, [p for p in ctx.params if p is not None])
        dx = kernels.linear_backward(x2, w, dy.reshape(-1, w.shape[0]), grads["w"], grads.get("b"), ctx.needs_input_grad[0], precision)
        ctx.params = None
        return (dx.view(shape) if dx is not None else None, ret[0], ret[1] if has_b else None, None)


class LstmStackFunction(torch.autograd.Function):
    """Stacked forward-running LSTM layers (nn.LSTM(num_layers=L, bidirectional=False)) on a packed batch, with optional
    initial state: (x (n_rows, In), h0 (L, n_seq, H) | None, c0 | None) -> (out (n_rows, H), h_n (L, n_seq, H), c_n)."""

    @staticmethod
    def forward(ctx, xp, sb, H, precision, h0, c0, *params):
        L = len(params) // 4
        acts, wss, hn, cn = [xp], [], [], []
        for l in range(L):
            h, hl, cl, ws = kernels.lstm_layer_forward(acts[-1], sb, params[4 * l:4 * l + 4], H,
                                                       None if h0 is None else h0[l].contiguous(),
                                                       None if c0 is None else c0[l].contiguous(), training=True, precision=precision)
            acts.append(h); wss.append(ws); hn.append(hl); cn.append(cl)
        ctx.meta = (sb, H, precision, L, h0 is not None, c0 is not None)
        ctx.wss, ctx.params = wss, params
        ctx.c0 = None if c0 is None else c0.detach()
        ctx.save_for_backward(*acts)
        return acts[-1], torch.stack(hn), torch.stack(cn)

    @staticmethod
    def backward(ctx, dout, dhn, dcn):
        sb, H, precision, L, has_h0, has_c0 = ctx.meta
        acts, params = ctx.saved_tensors, ctx.params
        grads, ret = _grad_targets([str(i) for i in range(len(params))], params)
        want_d0 = (has_h0 and ctx.needs_input_grad[4]) or (has_c0 and ctx.needs_input_grad[5])
        dh, dh0s, dc0s = dout, [], []
        for l in range(L - 1, -1, -1):
            want_dx = l > 0 or ctx.needs_input_grad[0]
            g4 = [grads[str(4 * l + i)] for i in range(4)]
            dh, dh0, dc0 = kernels.lstm_layer_backward(acts[l], acts[l + 1], dh, None if dhn is None else dhn[l],
                                                       None if dcn is None else dcn[l], sb, params[4 * l:4 * l + 4],
                                                       None if ctx.c0 is None else ctx.c0[l].contiguous(), g4, H, ctx.wss[l],
                                                       want_dx, want_d0, precision)
            dh0s.append(dh0); dc0s.append(dc0)
        ctx.wss = ctx.params = None
        gx = dh if ctx.needs_input_grad[0] else None
        gh0 = torch.stack(dh0s[::-1]) if (has_h0 and ctx.needs_input_grad[4]) else None
        gc0 = torch.stack(dc0s[::-1]) if (has_c0 and ctx.needs_input_grad[5]) else None
        return (gx, None, None, None, gh0, gc0) + tuple(ret)


class FrameHeadFunction(torch.autograd.Function):
    """probs = sigmoid(h w^T + b) for h (n, F): the Linear(F,1)+Sigmoid heads (cLSTM.out, sumgan.py:195-197,209)."""

    @staticmethod
    def forward(ctx, h, w, b):
        h = h.contiguous()
        s = kernels.frame_head_forward(h, w, b)
        ctx.save_for_backward(h, s, w)
        ctx.params = (w, b)
        return s

    @staticmethod
    def backward(ctx, ds):
        h, s, w = ctx.saved_tensors
        grads, ret = _grad_targets(["w", "b"], ctx.params)
        dh = kernels.frame_head_backward(h, s, ds, w, grads["w"], grads["b"])
        ctx.params = None
        return dh, ret[0], ret[1]


class LstmDecoderFunction(torch.autograd.Function):
    """SumGAN's step-wise dLSTM loop (sumgan.py:98-115) as one op: (h0, c0 (L, n_seq, H), *params) -> top-layer outputs
    (n_rows, H) in time order."""

    @staticmethod
    def forward(ctx, sb, H, h0, c0, *params):
        L = len(params) // 4
        layers = [params[4 * l:4 * l + 4] for l in range(L)]
        h0c = None if h0 is None else h0.contiguous()
        c0c = None if c0 is None else c0.contiguous()
        out, ws = kernels.lstm_decoder_forward(sb, layers, H, h0c, c0c)
        ctx.meta = (sb, H, L, h0 is not None, c0 is not None)
        ctx.ws, ctx.params, ctx.c0 = ws, params, (None if c0c is None else c0c.detach())
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        sb, H, L, has_h0, has_c0 = ctx.meta
        (out,) = ctx.saved_tensors
        params = ctx.params
        grads, ret = _grad_targets([str(i) for i in range(len(params))], params)
        want_d0 = (has_h0 and ctx.needs_input_grad[2]) or (has_c0 and ctx.needs_input_grad[3])
        dh0, dc0 = kernels.lstm_decoder_backward(sb, [params[4 * l:4 * l + 4] for l in range(L)], H, ctx.c0, out, dout,
                                                 [[grads[str(4 * l + i)] for i in range(4)] for l in range(L)], ctx.ws, want_d0)
        ctx.ws = ctx.params = None
        return (None, None, dh0 if (has_h0 and ctx.needs_input_grad[2]) else None,
                dc0 if (has_c0 and ctx.needs_input_grad[3]) else None) + tuple(ret)


class PolicyLossFunction(torch.autograd.Function):
    """DSNTrainer's REINFORCE loss glue for a packed batch in two HIP kernels (sumk_dsn_policy_loss_*, dsn.py:113-140):
    (probs (n_rows,), actions (E, n_rows), rewards (E, n_seq), base (n_seq,)) -> loss per video (n_seq,), already divided by E.
    Only `probs` receives a gradient (rewards and baselines are constants of the step, as in the reference)."""

    @staticmethod
    def forward(ctx, probs, sb, actions, rewards, base, beta, eps_target):
        probs_c, actions_c = probs.contiguous(), actions.contiguous()
        rewards_c, base_c = rewards.detach().contiguous(), base.detach().contiguous().float()
        lv, mp = kernels.dsn_policy_loss_forward(probs_c, sb, actions_c, rewards_c, base_c, beta, eps_target)
        ctx.save_for_backward(probs_c, actions_c, rewards_c, base_c, mp)
        ctx.meta = (sb, beta, eps_target)
        return lv

    @staticmethod
    def backward(ctx, dlv):
        probs, actions, rewards, base, mp = ctx.saved_tensors
        sb, beta, eps_target = ctx.meta
        dprobs = kernels.dsn_policy_loss_backward(probs, sb, actions, rewards, base, mp, dlv, beta, eps_target)
        return dprobs, None, None, None, None, None, None


class SegmentMseFunction(torch.autograd.Function):
    """Per-video nn.MSELoss of a packed batch in two HIP kernels (sumk_segment_mse_*): (scores (n_rows,), target (n_rows,)) ->
    (n_seq,).  Only `scores` receives a gradient."""

    @staticmethod
    def forward(ctx, scores, target, sb):
        from . import _lib
        lib = _lib.load()
        s, y = scores.contiguous(), target.detach().contiguous().float()
        kernels._require_gpu(s, "segment_mse scores"); kernels._require_gpu(y, "segment_mse target")
        if s.shape != (sb.n_rows,) or y.shape != (sb.n_rows,):
            raise kernels.SumkError(f"segment_mse: scores {tuple(s.shape)} / target {tuple(y.shape)} do not fit {sb.n_rows} rows")
        out = torch.empty(sb.n_seq, dtype=torch.float32, device=s.device)
        _lib.check(lib.sumk_segment_mse_forward(kernels._p(s), kernels._p(y), sb.n_seq, sb.off_dev_p, kernels._p(out), kernels._stream()),
                   "sumk_segment_mse_forward")
        ctx.save_for_backward(s, y)
        ctx.sb = sb
        return out

    @staticmethod
    def backward(ctx, dmse):
        from . import _lib
        s, y = ctx.saved_tensors
        ds = torch.empty_like(s)
        _lib.check(_lib.load().sumk_segment_mse_backward(kernels._p(s), kernels._p(y), kernels._p(dmse.contiguous()), ctx.sb.n_seq,
                                                         ctx.sb.off_dev_p, kernels._p(ds), kernels._stream()), "sumk_segment_mse_backward")
        return ds, None, None


class SegmentMseMeanFunction(torch.autograd.Function):
    """scale * sum over the videos of a packed batch of nn.MSELoss per video, as ONE scalar in one HIP kernel each way
    (sumk_segment_mse_mean_*): the trainers' step loss (vasnet.py:209-212; scale = 1 / videos of the step).  The same values and
    gradients as `SegmentMseFunction.apply(...).sum() * scale` without torch's reduction / expand kernels around it."""

    @staticmethod
    def forward(ctx, scores, target, sb, scale):
        from . import _lib
        lib = _lib.load()
        s, y = scores.contiguous(), target.detach().contiguous().float()
        kernels._require_gpu(s, "segment_mse scores"); kernels._require_gpu(y, "segment_mse target")
        if s.shape != (sb.n_rows,) or y.shape != (sb.n_rows,):
            raise kernels.SumkError(f"segment_mse: scores {tuple(s.shape)} / target {tuple(y.shape)} do not fit {sb.n_rows} rows")
        out = torch.empty(sb.n_seq + 1, dtype=torch.float32, device=s.device)       # [per-video ..., loss]
        ticket = getattr(sb, "_mse_ticket", None)                                   # one word per batch geometry, zeroed once; every launch leaves it zero
        if ticket is None:
            ticket = sb._mse_ticket = torch.zeros(1, dtype=torch.int32, device=s.device)
        _lib.check(lib.sumk_segment_mse_mean_forward(kernels._p(s), kernels._p(y), sb.n_seq, sb.off_dev_p, float(scale), kernels._p(out),
                                                     out.data_ptr() + 4 * sb.n_seq, kernels._p(ticket), kernels._stream()),
                   "sumk_segment_mse_mean_forward")
        ctx.save_for_backward(s, y)
        ctx.sb, ctx.scale = sb, float(scale)
        return out[sb.n_seq]

    @staticmethod
    def backward(ctx, dloss):
        from . import _lib
        s, y = ctx.saved_tensors
        ds = torch.empty_like(s)
        g = dloss.contiguous().float()
        _lib.check(_lib.load().sumk_segment_mse_mean_backward(kernels._p(s), kernels._p(y), kernels._p(g), ctx.scale, ctx.sb.n_seq,
                                                              ctx.sb.off_dev_p, kernels._p(ds), kernels._stream()), "sumk_segment_mse_mean_backward")
        return ds, None, None, None
