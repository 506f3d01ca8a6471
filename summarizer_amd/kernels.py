"""Thin Python layer over the C ABI: packed-batch bookkeeping (sequence offsets, workspaces) and
torch.autograd.Function wrappers.  Every arithmetic result comes from libsumk.so; nothing here computes on
the CPU or through torch ops (torch only allocates the buffers and carries the stream)."""
import ctypes as C
import numpy as np
import torch

from . import _lib
from ._lib import SumkError


def _require_gpu(t, what):
    if not t.is_cuda:
        raise SumkError(f"{what}: expected a GPU tensor, got device={t.device}. summarizer_amd runs only on the "
                        "HIP path (no CPU fallback); move the model and inputs to cuda.")
    if t.dtype != torch.float32:
        raise SumkError(f"{what}: expected float32, got {t.dtype}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SeqBatch:
    """Packed batch geometry: frames of video s are rows [off[s], off[s+1])."""
    _cache = {}

    def __init__(self, lens, device):
        lens = [int(v) for v in lens]
        if len(lens) == 0 or min(lens) <= 0:
            raise SumkError(f"empty video in batch: lens={lens}")
        self.lens = lens
        self.off_host = np.zeros(len(lens) + 1, dtype=np.int32)
        np.cumsum(lens, out=self.off_host[1:])
        self.n_seq = len(lens)
        self.n_rows = int(self.off_host[-1])
        self.device = device
        self.off_dev = torch.from_numpy(self.off_host).to(device)

    @classmethod
    def get(cls, lens, device):
        key = (tuple(int(v) for v in lens), str(device))
        sb = cls._cache.get(key)
        if sb is None:
            if len(cls._cache) > 256:
                cls._cache.clear()
            sb = cls._cache[key] = SeqBatch(lens, device)
        return sb

    def segment_mean(self, v):
        """Per-video mean of a per-frame quantity: v (..., n_rows) -> (..., n_seq).  Two tiny torch ops (index_add + div)
        whatever the number of videos -- the trainers' loss glue, not a compute kernel."""
        if getattr(self, "_seg", None) is None:
            self._seg = torch.repeat_interleave(torch.arange(self.n_seq, device=self.device),
                                                torch.tensor(self.lens, device=self.device))
            self._len_f = torch.tensor(self.lens, dtype=torch.float32, device=self.device)
        out = torch.zeros(v.shape[:-1] + (self.n_seq,), dtype=v.dtype, device=v.device)
        return out.index_add_(v.dim() - 1, self._seg, v) / self._len_f

    @property
    def off_host_p(self):
        return _lib.host_i32(self.off_host)

    @property
    def off_dev_p(self):
        return C.c_void_p(self.off_dev.data_ptr())


_ws_cache = {}


def workspace(nbytes, device, persistent=False):
    """Caller-owned scratch for one call.  Inference reuses one grow-only buffer per (device, stream) -- calls on one stream
    are ordered, calls on different streams (e.g. ingest.StreamingScorer's compute stream next to the default one) must not
    share scratch; a training forward gets its own buffer (it must survive until backward)."""
    if persistent:
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _ws_cache[key] = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
    return buf


_ONES = {}


def one(device):
    """A cached float32 scalar 1.0 on `device`: `loss.backward(gradient=kernels.one(dev))` spares autograd's ones_like + fill launch in
    every training step (the root gradient of a scalar loss)."""
    key = str(device)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ------------------------------------------------------------------------------------------------ VASNet
VASNET_FIELDS = [("Wk", "K.weight"), ("Wq", "Q.weight"), ("Wv", "V.weight"), ("Wo", "attention_head_projection.weight"),
                 ("W1", "k1.weight"), ("b1", "k1.bias"), ("w2", "k2.weight"), ("b2", "k2.bias"),
                 ("ln_w", "layer_norm.weight"), ("ln_b", "layer_norm.bias")]


PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16x6": 2, "bf16": 3}      # SUMK_PRECISION_* of include/sumk.h


def precision_code(p):
    """"fp32" (default: exact fp32 MFMA), "bf16x6" (fp32-grade, 6 bf16 MFMAs per product), "bf16x3" (hi+lo bf16 split, 3 MFMAs)
    or "bf16" (plain bf16 operands, 1 MFMA, fp32 accumulate: mixed-precision TRAINING arithmetic, ~2^-9 per product)."""
    if p is None:
        return 0
    if p not in PRECISIONS:
        raise SumkError(f"unknown precision {p!r}; expected one of {sorted(PRECISIONS)}")
    return PRECISIONS[p]


def vasnet_tables(sb, D, training, precision_id):
    """The batch's problem tables (sumk_vasnet_opts.tables), built once per (SeqBatch, D, training, arithmetic) and kept with the
    SeqBatch: they depend on the batch geometry only, so no table-setup kernel runs per call.  One buffer per CURRENT stream (the tables
    hold the tickets of the in-launch split-K launches: calls on different streams must not share them)."""
    key = (int(D), int(bool(training)), int(precision_id), torch.cuda.current_stream(sb.device).cuda_stream)
    tabs = getattr(sb, "_vasnet_tables", None)
    if tabs is None:
        tabs = sb._vasnet_tables = {}
    t = tabs.get(key)
    if t is None:
        lib = _lib.load()
        nb = lib.sumk_vasnet_tables_bytes(int(D), sb.n_seq, sb.off_host_p)
        if nb == 0:
            _lib.check(-1, "sumk_vasnet_tables_bytes")
        t = torch.empty(nb + 256, dtype=torch.uint8, device=sb.device)
        base = (t.data_ptr() + 255) // 256 * 256
        _lib.check(lib.sumk_vasnet_build_tables(int(D), sb.n_seq, sb.off_host_p, sb.off_dev_p, key[1], key[2], C.c_void_p(base), nb, _stream()),
                   "sumk_vasnet_build_tables")
        t = tabs[key] = (t, base)
    return t[1]


def _vasnet_structs(params, opts):
    w = _lib.VasnetWeights()
    for f, k in VASNET_FIELDS:
        t = params[k]
        _require_gpu(t, f"VASNet weight {k}")
        if not t.is_contiguous():
            raise SumkError(f"VASNet weight {k} must be contiguous")
        setattr(w, f, t.data_ptr())
    o = _lib.VasnetOpts(float(opts["scale"]), float(opts["eps"]), int(bool(opts.get("ignore_self", False))),
                        -1 if opts.get("aperture") is None else int(opts["aperture"]),
                        float(opts.get("dropout_p", 0.0)), int(opts.get("seed", 0)), precision_code(opts.get("precision")),
                        opts["seed_dev"].data_ptr() if opts.get("seed_dev") is not None else None, opts.get("tables"),
                        opts["x16"].data_ptr() if opts.get("x16") is not None else None,
                        opts["xplanes"].data_ptr() if opts.get("xplanes") is not None else None,
                        opts["wplanes"].data_ptr() if opts.get("wplanes") is not None else None)
    return w, o


def vasnet_wplanes(params, D, n_planes, wvo=None, out=None):
    """The weight-plane block of the plane path (sumk_vasnet_opts.wplanes) for the current weights: planes of [Wq; Wk; Wv] (or Wvo),
    Wo, W1 diag(ln_w) and the fused tail's column vectors.  Returns a 256-byte aligned uint8 tensor, or None when D is not eligible."""
    lib = _lib.load()
    nb = lib.sumk_vasnet_wplanes_bytes(int(D), int(n_planes))
    if nb == 0:
        return None
    w = _lib.VasnetWeights()
    for f, k in VASNET_FIELDS:
        setattr(w, f, params[k].data_ptr())
    dev = params[VASNET_FIELDS[0][1]].device
    if out is None or out.numel() < nb + 256 or out.device != dev:
        out = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    base = (out.data_ptr() + 255) // 256 * 256
    _lib.check(lib.sumk_vasnet_wplanes_build(int(D), C.byref(w), _p(wvo), int(n_planes), C.c_void_p(base), nb, _stream()), "sumk_vasnet_wplanes_build")
    view = out[base - out.data_ptr():]
    view._sumk_keep = out
    return view


def tensor_shadow(x, name, build):
    """A value derived from the CONTENTS of tensor x (a bf16 copy, bf16 planes ...), kept ON the tensor object: `build()` runs again when
    x was written to through torch (tensor version counter) and the entry dies with x.  Nothing is keyed by address: a fresh tensor that
    the caching allocator places where an earlier one lived (torch.cat of a new mini-batch every step) has no shadow and gets its own
    (the address-keyed cache handed such a tensor the previous batch's copy).  Writes torch cannot see (a C-ABI kernel
    writing into x) must be followed by `drop_shadows(x)`."""
    d = getattr(x, "_sumk_shadows", None)
    if d is None:
        d = {}
        try:
            x._sumk_shadows = d
        except AttributeError:          # (an object that takes no attributes: no caching)
            return build()
    # a shadow is built by kernels on the CURRENT stream: a call on another stream builds its own rather than read one that may not be finished
    stream = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0
    hit = d.get(name)
    if hit is not None and hit[0] == x._version and hit[2] == stream:
        return hit[1]
    val = build()
    d[name] = (x._version, val, stream)
    return val


def drop_shadows(x):
    """Forget every shadow of x (after a write torch's version counter does not see)."""
    if getattr(x, "_sumk_shadows", None):
        x._sumk_shadows.clear()


def vasnet_x16(x, sb=None):
    """bf16(x) for the mixed-precision training step (sumk_vasnet_opts.x16): kept with the tensor OBJECT while its contents are unchanged
    (tensor_shadow) -- features packed once are constant over the epochs of a run; a loop that packs a fresh batch every step converts per
    call as before."""
    lib = _lib.load()

    def build():
        x16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        _lib.check(lib.sumk_cast_f32_bf16(_p(x), _p(x16), x.numel(), _stream()), "sumk_cast_f32_bf16")
        return x16
    return tensor_shadow(x, "x16", build)


# ------------------------------------------------------------------------------------------------ KB planes (csrc/gemm_pw.hip)
PLANES_OF = {"bf16x6": 3, "bf16x3": 2}      # planes per operand of the split-bf16 arithmetics


def split_planes(x, n_planes):
    """fp32 (rows, K) row-major GPU tensor -> its KB-plane array (uint8 tensor; layout: include/sumk.h "KB planes")."""
    lib = _lib.load()
    _require_gpu(x, "split_planes")
    if x.dim() != 2 or x.stride(1) != 1 or x.stride(0) % 4 != 0:
        raise SumkError("split_planes: a 2-D row-major tensor with a leading dimension that is a multiple of 4")
    rows, K = x.shape
    nb = lib.sumk_planes_bytes(rows, K, n_planes)
    if nb == 0:
        raise SumkError(f"split_planes: rows={rows} K={K} planes={n_planes} is not representable (K % 16, 2 or 3 planes)")
    out = torch.empty(nb, dtype=torch.uint8, device=x.device)
    out[nb - 8192:].zero_()          # (the kernel writes every row up to the pitch, zeros behind `rows`; only the slack behind the last sub-array,
    _lib.check(lib.sumk_split_planes(_p(x), rows, K, x.stride(0), n_planes, _p(out), _stream()), "sumk_split_planes")      #  which row tiles read past, needs a fill)
    return out


def gemm_planes(a_planes, a_rows, b_planes, b_rows, M, N, K, n_planes, variant=0, out=None):
    """C (M, N) fp32 = A . B^T from two KB-plane arrays (tests / bench probe of the plane-aware wide GEMM)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a_planes.device)
    _lib.check(lib.sumk_gemm_planes(_p(a_planes), a_rows, _p(b_planes), b_rows, _p(out), M, N, K, n_planes, variant, _stream()), "sumk_gemm_planes")
    return out


def fold_vo(w_o, w_v, out=None):
    """Wvo = Wo . Wv (D, D) for the folded inference path (sumk_vasnet_forward_folded); one fp32 MFMA GEMM."""
    lib = _lib.load()
    _require_gpu(w_o, "fold_vo"); _require_gpu(w_v, "fold_vo")
    D = w_o.shape[0]
    if out is None:
        out = torch.empty(D, D, dtype=torch.float32, device=w_o.device)
    _lib.check(lib.sumk_gemm_nn(_p(w_o.contiguous()), _p(w_v.contiguous()), _p(out), D, D, D, _stream()), "sumk_gemm_nn (fold_vo)")
    return out


def vasnet_forward_packed(x, sb, params, opts, pos_table=None, pos_rows=None, training=False, wvo=None):
    """x: (n_rows, D) packed, contiguous, on GPU.  Returns (scores (n_rows,), workspace or None).
    wvo: folded Wo.Wv (inference only) -> the output projection is absorbed into the packed projection (fold_vo)."""
    lib = _lib.load()
    _require_gpu(x, "vasnet input")
    if not x.is_contiguous() or x.dim() != 2 or x.shape[0] != sb.n_rows:
        raise SumkError(f"vasnet input must be contiguous (n_rows={sb.n_rows}, D), got {tuple(x.shape)}")
    D = x.shape[1]
    # Entries this function derives (tables of the batch geometry, the bf16 / plane shadows of x) are written back into `opts` so that the
    # backward pass of the same step finds them; they are tied to THIS (x, batch): a caller that re-uses the dict for another batch gets
    # them rebuilt instead of silently scoring with the first batch's planes (sumk.h: stale planes give wrong results, not errors).
    dkey = (x.data_ptr(), x._version, tuple(x.shape), id(sb))
    if opts.get("_derived_for") != dkey:
        for k in opts.pop("_derived", ()):
            opts.pop(k, None)
        opts["_derived_for"], opts["_derived"] = dkey, []
    if "tables" not in opts and not torch.cuda.is_current_stream_capturing():      # (built outside a capture; a captured call reuses what exists)
        opts["tables"] = vasnet_tables(sb, D, training, precision_code(opts.get("precision"))); opts["_derived"].append("tables")
    elif "tables" not in opts:
        # inside a capture nothing may be allocated or built for the (private) capture stream: the step reuses the tables an eager call
        # of the same geometry built before (a replay runs where the eager calls ran); none yet -> the setup kernel is captured instead
        want = (int(D), int(bool(training)), precision_code(opts.get("precision")))
        hit = [v for k, v in getattr(sb, "_vasnet_tables", {}).items() if k[:3] == want]
        opts["tables"] = hit[-1][1] if hit else None; opts["_derived"].append("tables")
    if (training and "x16" not in opts and precision_code(opts.get("precision")) == precision_code("bf16") and pos_table is None
            and x.numel() % 4 == 0 and not x.requires_grad and not torch.cuda.is_current_stream_capturing()):
        opts["x16"] = vasnet_x16(x, sb); opts["_derived"].append("x16")        # (an input that asks for dX is an activation, not a dataset: cast per call)
    n_planes = PLANES_OF.get(opts.get("precision"))
    if (n_planes and not training and "xplanes" not in opts and opts.get("wplanes") is not None and pos_table is None
            and D % 256 == 0 and sb.n_rows >= 256 and not torch.cuda.is_current_stream_capturing()):
        # plane path (csrc/gemm_pw.hip): the operand planes of x are kept with the tensor object (constant per dataset)
        opts["xplanes"] = tensor_shadow(x, f"planes{n_planes}", lambda: split_planes(x, n_planes)); opts["_derived"].append("xplanes")
    w, o = _vasnet_structs(params, opts)
    nbytes = lib.sumk_vasnet_workspace_bytes_for(D, sb.n_seq, sb.off_host_p, int(training), int(o.precision))
    if nbytes == 0:
        _lib.check(-1, "sumk_vasnet_workspace_bytes_for")
    ws = workspace(nbytes, x.device, persistent=training)
    scores = torch.empty(sb.n_rows, dtype=torch.float32, device=x.device)
    if wvo is not None:
        if training:
            raise SumkError("vasnet: the folded projection is inference-only")
        rc = lib.sumk_vasnet_forward_folded(_p(x), D, sb.n_seq, sb.off_host_p, sb.off_dev_p, C.byref(w), _p(wvo), C.byref(o),
                                            _p(pos_table), _p(pos_rows), _p(scores), _p(ws), ws.numel(), _stream())
    else:
        rc = lib.sumk_vasnet_forward(_p(x), D, sb.n_seq, sb.off_host_p, sb.off_dev_p, C.byref(w), C.byref(o),
                                     _p(pos_table), _p(pos_rows), _p(scores), _p(ws), ws.numel(), int(training), _stream())
    _lib.check(rc, "sumk_vasnet_forward")
    return scores, (ws if training else None)


# ------------------------------------------------------------------------------------------------ BiLSTM scorers
# SUMK_CHECK=1 (set by the test-suite): synchronise after every recurrent layer and verify that the persistent kernel's
# bounded waits did not time out.  Off by default (it costs a host sync per layer).
import os as _os
CHECK_LSTM = _os.environ.get("SUMK_CHECK", "0") == "1"


def _lstm_layer_struct(params, prefix, layer):
    w = _lib.LstmLayerWeights()
    for d, suf in enumerate(("", "_reverse")):
        for f, n in (("w_ih", "weight_ih"), ("w_hh", "weight_hh"), ("b_ih", "bias_ih"), ("b_hh", "bias_hh")):
            t = params[f"{prefix}{n}_l{layer}{suf}"]
            _require_gpu(t, f"LSTM weight {prefix}{n}_l{layer}{suf}")
            if not t.is_contiguous():
                raise SumkError(f"LSTM weight {prefix}{n}_l{layer}{suf} must be contiguous")
            getattr(w, f)[d] = t.data_ptr()
    return w


def bilstm_wplanes(params, prefix, layer, In, H, n_planes, out=None):
    """Weight-plane block of one BiLSTM layer's input projection (sumk_lstm_layer_weights.w_planes), or None when the shape is not eligible."""
    lib = _lib.load()
    nb = lib.sumk_bilstm_wplanes_bytes(int(In), int(H), int(n_planes))
    if nb == 0:
        return None
    w = _lstm_layer_struct(params, prefix, layer)
    dev = params[f"{prefix}weight_ih_l{layer}"].device
    if out is None or out.numel() < nb + 256 or out.device != dev:
        out = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    base = (out.data_ptr() + 255) // 256 * 256
    _lib.check(lib.sumk_bilstm_wplanes_build(int(In), int(H), C.byref(w), int(n_planes), C.c_void_p(base), nb, _stream()), "sumk_bilstm_wplanes_build")
    view = out[base - out.data_ptr():]
    view._sumk_keep = out
    return view


def bilstm_layer_forward(x, sb, params, prefix, layer, H, training=False, precision=None, wplanes=None, dataset_input=False):
    """x: (n_rows, In) packed -> h (n_rows, 2H) = [h_fwd || h_rev].  Returns (h, workspace or None).
    wplanes (inference in bf16x6 / bf16x3): the layer's weight-plane block -- the input projection then runs on operand planes
    (csrc/gemm_pw.hip); x's planes are kept with the tensor object when dataset_input (layer 0: features), split per call otherwise."""
    lib = _lib.load()
    _require_gpu(x, "bilstm input")
    if not x.is_contiguous() or x.dim() != 2 or x.shape[0] != sb.n_rows:
        raise SumkError(f"bilstm input must be contiguous (n_rows={sb.n_rows}, In), got {tuple(x.shape)}")
    In = x.shape[1]
    w = _lstm_layer_struct(params, prefix, layer)
    n_planes = PLANES_OF.get(precision)
    xplanes = None
    if wplanes is not None and n_planes and not training and sb.n_rows >= 1024:
        xplanes = (tensor_shadow(x, f"planes{n_planes}", lambda: split_planes(x, n_planes)) if dataset_input and not torch.cuda.is_current_stream_capturing()
                   else split_planes(x, n_planes))
        w.x_planes, w.w_planes = xplanes.data_ptr(), wplanes.data_ptr()
    nbytes = lib.sumk_bilstm_workspace_bytes(In, H, sb.n_seq, sb.off_host_p, int(training))
    if nbytes == 0:
        _lib.check(-1, "sumk_bilstm_workspace_bytes")
    ws = workspace(nbytes, x.device, persistent=training)
    h = torch.empty(sb.n_rows, 2 * H, dtype=torch.float32, device=x.device)
    rc = lib.sumk_bilstm_layer_forward(_p(x), In, H, sb.n_seq, sb.off_host_p, sb.off_dev_p, C.byref(w), _p(h), _p(ws),
                                       ws.numel(), int(training), precision_code(precision), _stream())
    _lib.check(rc, "sumk_bilstm_layer_forward")
    if CHECK_LSTM:
        _lib.check(lib.sumk_bilstm_check(_p(ws), In, H, sb.n_seq, sb.off_host_p, int(training), 0, _stream()), "sumk_bilstm_check")
    return h, (ws if training else None)


def health_check():
    """Raises SumkError if a persistent recurrence kernel on the current device timed out since the last check (its
    outputs, and everything computed from them, are invalid).  Synchronises the current stream; meant for the host
    synchronisation points callers already have (score D2H, per-epoch loss read-back)."""
    _lib.check(_lib.load().sumk_health_check(_stream()), "sumk_health_check")


def mfma_sustained_rate(kind="bf16", iters=4000, random_operands=False):
    """Measurement utility: dense TFLOP/s the current device sustains with nothing but MFMAs in flight (csrc/mfma_probe.hip) -- the practical ceiling of
    an MFMA-bound launch on THIS box, beside the guide's peak.  kind: "bf16" (32x32x16), "f32" (32x32x2), "bf16_16" (16x16x32).  Returns (tflops, seconds)."""
    code = {"bf16": 0, "f32": 1, "bf16_16": 2}[kind] + (4 if random_operands else 0)
    t, sec = C.c_double(0.0), C.c_double(0.0)
    _lib.check(_lib.load().sumk_probe_mfma_rate(code, int(iters), C.byref(t), C.byref(sec), _stream()), "sumk_probe_mfma_rate")
    return t.value, sec.value


def frame_head_forward(h, w, b):
    """scores = sigmoid(h @ w.T + b) for h (n_rows, F), w (1, F) or (F,), b (1,)."""
    lib = _lib.load()
    _require_gpu(h, "frame head input")
    scores = torch.empty(h.shape[0], dtype=torch.float32, device=h.device)
    rc = lib.sumk_frame_head_forward(_p(h), h.shape[0], h.shape[1], _p(w), _p(b), _p(scores), _stream())
    _lib.check(rc, "sumk_frame_head_forward")
    return scores


# ------------------------------------------------------------------------------------------------ training
def vasnet_backward_packed(x, sb, params, opts, dscores, ws, grads, want_dx=False):
    """Accumulates d(sum dscores*scores)/d(weights) into `grads` (dict state_dict-key -> preallocated tensor)."""
    lib = _lib.load()
    D = x.shape[1]
    w, o = _vasnet_structs(params, opts)
    g = _lib.VasnetGrads()
    for f, k in VASNET_FIELDS:
        t = grads[k]
        _require_gpu(t, f"VASNet grad {k}")
        setattr(g, f, t.data_ptr())
    dx = torch.empty_like(x) if want_dx else None
    if not dscores.is_contiguous():
        dscores = dscores.contiguous()
    ev = opts.get("tail_grads_ready_event")          # torch.cuda.Event recorded mid-backward (data-parallel overlap) or None
    if ev is not None:
        ev.record()                                  # materialise the lazy hipEvent_t handle (re-recorded by the library)
    rc = lib.sumk_vasnet_backward(_p(x), D, sb.n_seq, sb.off_host_p, sb.off_dev_p, C.byref(w), C.byref(o), _p(dscores),
                                  C.byref(g), _p(dx), _p(ws), ws.numel(), _stream(),
                                  C.c_void_p(ev.cuda_event) if ev is not None else C.c_void_p(0))
    _lib.check(rc, "sumk_vasnet_backward")
    return dx


# Bumped by every C-ABI call that WRITES model weights (the optimiser kernels): torch cannot see those writes (no ._version bump,
# same storage), so caches derived from weights -- VASNet's folded Wvo -- carry this counter in their key.
WEIGHTS_EPOCH = [0]


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """In-place torch.optim.Adam-equivalent update of one flat fp32 buffer (HIP kernel)."""
    lib = _lib.load()
    for t in (param, grad, exp_avg, exp_avg_sq):
        _require_gpu(t, "adam_step")
        if not t.is_contiguous():
            raise SumkError("adam_step: buffers must be contiguous")
    rc = lib.sumk_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0], betas[1], eps,
                            weight_decay, int(step), float(grad_scale), _stream())
    _lib.check(rc, "sumk_adam_step")
    WEIGHTS_EPOCH[0] += 1


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0,
                  sumsq=None, max_norm=0.0, zero_grad=False):
    """The same update with the step counter (state[0], int32, incremented by the call) and, when `sumsq` (device scalar, squared
    L2 norm of the unscaled gradient) is given, the clip_grad_norm_ coefficient kept ON THE DEVICE: no host sync, capturable into
    a HIP graph.  `state`: 4 x int32 device block, zeroed once by the owner."""
    lib = _lib.load()
    for t in (param, grad, exp_avg, exp_avg_sq):
        _require_gpu(t, "adam_step_dev")
        if not t.is_contiguous():
            raise SumkError("adam_step_dev: buffers must be contiguous")
    if not state.is_cuda or state.dtype != torch.int32 or state.numel() < 4 or not state.is_contiguous():
        raise SumkError("adam_step_dev: state must be a contiguous int32 GPU tensor of 4 elements")
    fn = lib.sumk_adam_step_dev_zero_grad if zero_grad else lib.sumk_adam_step_dev      # zero_grad: the gradient buffer is left zero
    rc = fn(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0], betas[1], eps,
            weight_decay, _p(state), float(grad_scale), None if sumsq is None else _p(sumsq), float(max_norm), _stream())
    _lib.check(rc, "sumk_adam_step_dev")
    WEIGHTS_EPOCH[0] += 1


def cast_f32_bf16(src, dst):
    """dst (bf16, same numel) <- src (fp32), round to nearest even (HIP)."""
    _require_gpu(src, "cast_f32_bf16")
    if dst.dtype != torch.bfloat16 or dst.numel() != src.numel() or not (src.is_contiguous() and dst.is_contiguous()):
        raise SumkError("cast_f32_bf16: dst must be a contiguous bfloat16 buffer of the same size")
    _lib.check(_lib.load().sumk_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "sumk_cast_f32_bf16")


def cast_bf16_f32(src, dst):
    """dst (fp32) <- src (bf16) (HIP)."""
    _require_gpu(dst, "cast_bf16_f32")
    if src.dtype != torch.bfloat16 or dst.numel() != src.numel() or not (src.is_contiguous() and dst.is_contiguous()):
        raise SumkError("cast_bf16_f32: src must be a contiguous bfloat16 buffer of the same size")
    _lib.check(_lib.load().sumk_cast_bf16_f32(_p(src), _p(dst), src.numel(), _stream()), "sumk_cast_bf16_f32")


_sumsq_ws = {}


def sumsq(v, out=None):
    """out[0] += sum(v*v) (HIP, deterministic).  Returns the 1-element device tensor."""
    lib = _lib.load()
    _require_gpu(v, "sumsq")
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=v.device)
    ws = _sumsq_ws.get(str(v.device))
    if ws is None:
        ws = _sumsq_ws[str(v.device)] = torch.empty(lib.sumk_sumsq_workspace_bytes(), dtype=torch.uint8, device=v.device)
    _lib.check(lib.sumk_sumsq(_p(v), v.numel(), _p(out), _p(ws), _stream()), "sumk_sumsq")
    return out


def bilstm_layer_backward(x, h, dh, sb, params, grads, prefix, layer, H, ws, want_dx, precision=None, tail_event=None):
    """BPTT of one bidirectional layer; accumulates into grads[<prefix>{weight,bias}_{ih,hh}_l{layer}[_reverse]].
    tail_event (torch.cuda.Event, data-parallel trainers): recorded by the library once the biases' and the reverse direction's gradients
    are final, before the forward direction's weight-gradient GEMMs (sumk_lstm_layer_grads.tail_ready_event)."""
    lib = _lib.load()
    In = x.shape[1]
    w = _lstm_layer_struct(params, prefix, layer)
    g = _lib.LstmLayerGrads()
    for d, suf in enumerate(("", "_reverse")):
        for f, n in (("w_ih", "weight_ih"), ("w_hh", "weight_hh"), ("b_ih", "bias_ih"), ("b_hh", "bias_hh")):
            getattr(g, f)[d] = grads[f"{prefix}{n}_l{layer}{suf}"].data_ptr()
    if tail_event is not None:
        tail_event.record()                          # materialise the lazy hipEvent_t handle (re-recorded by the library)
        g.tail_ready_event = tail_event.cuda_event
    dx = torch.empty_like(x) if want_dx else None
    if not dh.is_contiguous():
        dh = dh.contiguous()
    rc = lib.sumk_bilstm_layer_backward(_p(x), _p(h), _p(dh), In, H, sb.n_seq, sb.off_host_p, sb.off_dev_p, C.byref(w),
                                        C.byref(g), _p(dx), _p(ws), ws.numel(), precision_code(precision), _stream())
    _lib.check(rc, "sumk_bilstm_layer_backward")
    if CHECK_LSTM:
        _lib.check(lib.sumk_bilstm_check(_p(ws), In, H, sb.n_seq, sb.off_host_p, 1, 1, _stream()), "sumk_bilstm_check")
    return dx


def frame_head_backward(h, scores, dscores, w, dw, db):
    lib = _lib.load()
    F = h.shape[1]
    nb = lib.sumk_frame_head_workspace_bytes(F)
    ws = torch.empty(nb, dtype=torch.uint8, device=h.device)
    dh = torch.empty_like(h)
    if not dscores.is_contiguous():
        dscores = dscores.contiguous()
    rc = lib.sumk_frame_head_backward(_p(h), _p(scores), _p(dscores), h.shape[0], F, _p(w), _p(dh), _p(dw), _p(db), _p(ws),
                                      nb, _stream())
    _lib.check(rc, "sumk_frame_head_backward")
    return dh


# ------------------------------------------------------------------------------------------------ DSN reward
def dsn_reward(x, sb, actions, far_sim=False, temp_dist_thre=20):
    """actions: (E, n_rows) 0/1 floats -> rewards (E, n_seq) (dsn.py:185-236) for a packed batch."""
    lib = _lib.load()
    _require_gpu(x, "dsn_reward features"); _require_gpu(actions, "dsn_reward actions")
    E = actions.shape[0]
    if actions.dim() != 2 or actions.shape[1] != sb.n_rows or not actions.is_contiguous():
        raise SumkError(f"dsn_reward: actions must be contiguous (E, {sb.n_rows}), got {tuple(actions.shape)}")
    D = x.shape[1]
    nb = lib.sumk_dsn_reward_workspace_bytes(D, sb.n_seq, sb.off_host_p, E)
    if nb == 0:
        _lib.check(-1, "sumk_dsn_reward_workspace_bytes")
    ws = workspace(nb, x.device)
    out = torch.empty(E, sb.n_seq, dtype=torch.float32, device=x.device)
    rc = lib.sumk_dsn_reward(_p(x), D, sb.n_seq, sb.off_host_p, sb.off_dev_p, _p(actions), E, int(bool(far_sim)),
                             int(temp_dist_thre), _p(out), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "sumk_dsn_reward")
    return out


def dsn_policy_loss_forward(probs, sb, actions, rewards, base, beta, eps_target):
    """(loss_per_video (n_seq,), mean_probs (n_seq,)) of sumk_dsn_policy_loss_forward (include/sumk.h)."""
    lib = _lib.load()
    for t, what in ((probs, "probs"), (actions, "actions"), (rewards, "rewards"), (base, "base")):
        _require_gpu(t, "dsn_policy_loss " + what)
        if not t.is_contiguous():
            raise SumkError(f"dsn_policy_loss: {what} must be contiguous")
    E = actions.shape[0]
    if probs.shape != (sb.n_rows,) or actions.shape != (E, sb.n_rows) or rewards.shape != (E, sb.n_seq) or base.shape != (sb.n_seq,):
        raise SumkError(f"dsn_policy_loss: shapes probs {tuple(probs.shape)} actions {tuple(actions.shape)} rewards "
                        f"{tuple(rewards.shape)} base {tuple(base.shape)} do not fit {sb.n_rows} rows / {sb.n_seq} videos")
    lv = torch.empty(sb.n_seq, dtype=torch.float32, device=probs.device)
    mp = torch.empty(sb.n_seq, dtype=torch.float32, device=probs.device)
    rc = lib.sumk_dsn_policy_loss_forward(_p(probs), _p(actions), _p(rewards), _p(base), sb.n_seq, sb.n_rows, sb.off_dev_p, E,
                                          float(beta), float(eps_target), _p(lv), _p(mp), _stream())
    _lib.check(rc, "sumk_dsn_policy_loss_forward")
    return lv, mp


def dsn_policy_loss_backward(probs, sb, actions, rewards, base, mean_probs, dlv, beta, eps_target):
    lib = _lib.load()
    dlv = dlv.contiguous()
    dprobs = torch.empty_like(probs)
    rc = lib.sumk_dsn_policy_loss_backward(_p(probs), _p(actions), _p(rewards), _p(base), _p(mean_probs), _p(dlv), sb.n_seq, sb.n_rows,
                                           sb.off_dev_p, actions.shape[0], float(beta), float(eps_target), _p(dprobs), _stream())
    _lib.check(rc, "sumk_dsn_policy_loss_backward")
    return dprobs


# ------------------------------------------------------------------------------------------------ Transformer scorer
TF_LAYER_FIELDS = (("in_proj_w", "self_attn.in_proj_weight"), ("in_proj_b", "self_attn.in_proj_bias"),
                   ("out_proj_w", "self_attn.out_proj.weight"), ("out_proj_b", "self_attn.out_proj.bias"),
                   ("lin1_w", "linear1.weight"), ("lin1_b", "linear1.bias"), ("lin2_w", "linear2.weight"),
                   ("lin2_b", "linear2.bias"), ("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"),
                   ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"))
TF_HEAD_FIELDS = (("ln_w", "layer_norm.weight"), ("ln_b", "layer_norm.bias"), ("k1_w", "k1.weight"), ("k1_b", "k1.bias"),
                  ("k2_w", "k2.weight"), ("k2_b", "k2.bias"))


def transformer_param_names(n_layers):
    """state_dict keys the HIP path reads (the prototype `transformer_encoder_layer.*` copy is never used)."""
    names = [f"transformer_encoder.layers.{l}.{k}" for l in range(n_layers) for _, k in TF_LAYER_FIELDS]
    return names + [k for _, k in TF_HEAD_FIELDS]


def _tf_structs(tensors, n_layers, what, layer_cls, head_cls):
    def ptr(k):
        t = tensors[k]
        _require_gpu(t, f"Transformer {what} {k}")
        if not t.is_contiguous():
            raise SumkError(f"Transformer {what} {k} must be contiguous")
        return t.data_ptr()
    layers = (layer_cls * n_layers)()
    for l in range(n_layers):
        for f, k in TF_LAYER_FIELDS:
            setattr(layers[l], f, ptr(f"transformer_encoder.layers.{l}.{k}"))
    head = head_cls(*[ptr(k) for _, k in TF_HEAD_FIELDS])
    return layers, head


def _tf_opts(o):
    wp = o.get("wplanes")
    return _lib.TfOpts(float(o["layer_eps"]), float(o["final_eps"]), int(bool(o.get("more_residuals", False))),
                       float(o.get("layer_dropout_p", 0.0)), float(o.get("head_dropout_p", 0.0)), int(o.get("seed", 0)),
                       precision_code(o.get("precision")), wp.data_ptr() if wp is not None else None)


def transformer_wplanes(params, D, dff, n_layers, n_planes, out=None):
    """The weight-plane block of the Transformer scorer's plane path (sumk_tf_opts.wplanes) for the current weights: planes of in_proj,
    out_proj, linear1, linear2 of every layer and of k1.  Returns a 256-byte aligned uint8 tensor, or None when (D, dff) is not eligible."""
    lib = _lib.load()
    nb = lib.sumk_transformer_wplanes_bytes(int(D), int(dff), int(n_layers), int(n_planes))
    if nb == 0:
        return None
    layers, head = _tf_structs(params, n_layers, "weight", _lib.TfLayerWeights, _lib.TfHeadWeights)
    dev = params[TF_HEAD_FIELDS[0][1]].device
    if out is None or out.numel() < nb + 256 or out.device != dev:
        out = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    base = (out.data_ptr() + 255) // 256 * 256
    _lib.check(lib.sumk_transformer_wplanes_build(int(D), int(dff), int(n_layers), C.cast(layers, C.c_void_p), C.cast(C.pointer(head), C.c_void_p),
                                                  int(n_planes), C.c_void_p(base), nb, _stream()), "sumk_transformer_wplanes_build")
    view = out[base - out.data_ptr():]
    view._sumk_keep = out
    return view


def transformer_forward_packed(x, sb, params, n_layers, n_heads, dff, opts, pos_table=None, pos_rows=None, training=False):
    """x: (n_rows, D) packed -> (scores (n_rows,), workspace or None).  params: state_dict-keyed tensors."""
    lib = _lib.load()
    _require_gpu(x, "transformer input")
    if not x.is_contiguous() or x.dim() != 2 or x.shape[0] != sb.n_rows:
        raise SumkError(f"transformer input must be contiguous (n_rows={sb.n_rows}, D), got {tuple(x.shape)}")
    D = x.shape[1]
    layers, head = _tf_structs(params, n_layers, "weight", _lib.TfLayerWeights, _lib.TfHeadWeights)
    o = _tf_opts(opts)
    nbytes = lib.sumk_transformer_workspace_bytes_for(D, dff, n_heads, n_layers, sb.n_seq, sb.off_host_p, int(training),
                                                      o.precision if o.wplanes else 0)
    if nbytes == 0:
        _lib.check(-1, "sumk_transformer_workspace_bytes_for")
    ws = workspace(nbytes, x.device, persistent=training)
    scores = torch.empty(sb.n_rows, dtype=torch.float32, device=x.device)
    rc = lib.sumk_transformer_forward(_p(x), D, dff, n_heads, n_layers, sb.n_seq, sb.off_host_p, sb.off_dev_p,
                                      C.cast(layers, C.c_void_p), C.cast(C.pointer(head), C.c_void_p),
                                      C.cast(C.pointer(o), C.c_void_p), _p(pos_table), _p(pos_rows), _p(scores), _p(ws),
                                      ws.numel(), int(training), _stream())
    _lib.check(rc, "sumk_transformer_forward")
    return scores, (ws if training else None)


def transformer_backward_packed(x, sb, params, grads, n_layers, n_heads, dff, opts, dscores, ws, want_dx=False):
    lib = _lib.load()
    D = x.shape[1]
    layers, head = _tf_structs(params, n_layers, "weight", _lib.TfLayerWeights, _lib.TfHeadWeights)
    glayers, ghead = _tf_structs(grads, n_layers, "grad", _lib.TfLayerWeights, _lib.TfHeadWeights)   # same field layout
    o = _tf_opts(opts)
    dx = torch.empty_like(x) if want_dx else None
    if not dscores.is_contiguous():
        dscores = dscores.contiguous()
    rc = lib.sumk_transformer_backward(_p(x), D, dff, n_heads, n_layers, sb.n_seq, sb.off_host_p, sb.off_dev_p,
                                       C.cast(layers, C.c_void_p), C.cast(C.pointer(head), C.c_void_p),
                                       C.cast(C.pointer(o), C.c_void_p), _p(dscores), C.cast(glayers, C.c_void_p),
                                       C.cast(C.pointer(ghead), C.c_void_p), _p(dx), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "sumk_transformer_backward")
    return dx


# ------------------------------------------------------------------------------------------------ unidirectional LSTM layer / dense layer
def _lstm_dir_struct(cls, tensors):
    s = cls()
    for f, t in zip(("w_ih", "w_hh", "b_ih", "b_hh"), tensors):
        _require_gpu(t, f"LSTM {f}")
        if not t.is_contiguous():
            raise SumkError(f"LSTM {f} must be contiguous")
        setattr(s, f, t.data_ptr())
    return s


def _state(t, n_seq, H, what):
    if t is None:
        return None
    _require_gpu(t, what)
    if tuple(t.shape) != (n_seq, H) or not t.is_contiguous():
        raise SumkError(f"{what} must be contiguous ({n_seq}, {H}), got {tuple(t.shape)}")
    return t


def lstm_layer_forward(x, sb, weights, H, h0=None, c0=None, training=False, precision=None):
    """One forward-running LSTM layer (nn.LSTM(bidirectional=False) semantics) on a packed batch.
    x (n_rows, In); weights = (w_ih (4H,In), w_hh (4H,H), b_ih, b_hh); h0 / c0 (n_seq, H) or None.
    Returns (h_out (n_rows, H), h_last (n_seq, H), c_last (n_seq, H), workspace or None)."""
    lib = _lib.load()
    _require_gpu(x, "lstm input")
    if not x.is_contiguous() or x.dim() != 2 or x.shape[0] != sb.n_rows:
        raise SumkError(f"lstm input must be contiguous (n_rows={sb.n_rows}, In), got {tuple(x.shape)}")
    In = x.shape[1]
    w = _lstm_dir_struct(_lib.LstmDirWeights, weights)
    h0, c0 = _state(h0, sb.n_seq, H, "h0"), _state(c0, sb.n_seq, H, "c0")
    nbytes = lib.sumk_lstm_workspace_bytes(In, H, sb.n_seq, sb.off_host_p, int(training))
    if nbytes == 0:
        _lib.check(-1, "sumk_lstm_workspace_bytes")
    ws = workspace(nbytes, x.device, persistent=training)
    h = torch.empty(sb.n_rows, H, dtype=torch.float32, device=x.device)
    h_last = torch.empty(sb.n_seq, H, dtype=torch.float32, device=x.device)
    c_last = torch.empty(sb.n_seq, H, dtype=torch.float32, device=x.device)
    rc = lib.sumk_lstm_layer_forward(_p(x), In, H, sb.n_seq, sb.off_host_p, sb.off_dev_p, C.byref(w), _p(h0), _p(c0), _p(h),
                                     _p(h_last), _p(c_last), _p(ws), ws.numel(), int(training), precision_code(precision), _stream())
    _lib.check(rc, "sumk_lstm_layer_forward")
    return h, h_last, c_last, (ws if training else None)


def lstm_layer_backward(x, h, dh, dh_last, dc_last, sb, weights, c0, grads, H, ws, want_dx, want_d0, precision=None):
    """BPTT of one forward-running layer; ACCUMULATES into grads = (dw_ih, dw_hh, db_ih, db_hh).  dh (n_rows, H), dh_last,
    dc_last (n_seq, H) may be None.  Returns (dx or None, dh0 or None, dc0 or None)."""
    lib = _lib.load()
    In = x.shape[1]
    w = _lstm_dir_struct(_lib.LstmDirWeights, weights)
    g = _lstm_dir_struct(_lib.LstmDirGrads, grads)
    dx = torch.empty_like(x) if want_dx else None
    dh0 = torch.empty(sb.n_seq, H, dtype=torch.float32, device=x.device) if want_d0 else None
    dc0 = torch.empty(sb.n_seq, H, dtype=torch.float32, device=x.device) if want_d0 else None
    dh = dh.contiguous() if dh is not None else None
    dh_last = dh_last.contiguous() if dh_last is not None else None
    dc_last = dc_last.contiguous() if dc_last is not None else None
    rc = lib.sumk_lstm_layer_backward(_p(x), _p(h), _p(dh), _p(dh_last), _p(dc_last), In, H, sb.n_seq, sb.off_host_p, sb.off_dev_p,
                                      C.byref(w), _p(c0), C.byref(g), _p(dx), _p(dh0), _p(dc0), _p(ws), ws.numel(),
                                      precision_code(precision), _stream())
    _lib.check(rc, "sumk_lstm_layer_backward")
    return dx, dh0, dc0


def _linear_ws(N, K, device):
    nb = _lib.load().sumk_linear_workspace_bytes(N, K)
    return torch.empty(nb, dtype=torch.uint8, device=device)


def linear_forward(x, w, b, precision=None):
    """y (M, N) = x (M, K) w^T + b, w (N, K) as in nn.Linear."""
    lib = _lib.load()
    _require_gpu(x, "linear input")
    x = x.contiguous()
    M, K = x.shape
    N = w.shape[0]
    ws = _linear_ws(N, K, x.device)
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    _lib.check(lib.sumk_linear_forward(_p(x), _p(w), _p(b), _p(y), M, N, K, _p(ws), ws.numel(), precision_code(precision), _stream()),
               "sumk_linear_forward")
    return y


def linear_backward(x, w, dy, dw, db, want_dx, precision=None):
    """dw (N, K) / db (N) ACCUMULATED if given; returns dx (M, K) or None."""
    lib = _lib.load()
    x, dy = x.contiguous(), dy.contiguous()
    M, K = x.shape
    N = w.shape[0]
    ws = _linear_ws(N, K, x.device)
    dx = torch.empty_like(x) if want_dx else None
    _lib.check(lib.sumk_linear_backward(_p(x), _p(w), _p(dy), M, N, K, _p(dx), _p(dw), _p(db), _p(ws), ws.numel(),
                                        precision_code(precision), _stream()), "sumk_linear_backward")
    return dx


def _dir_array(cls, per_layer):
    arr = (cls * len(per_layer))()
    for l, tensors in enumerate(per_layer):
        for f, t in zip(("w_ih", "w_hh", "b_ih", "b_hh"), tensors):
            _require_gpu(t, f"decoder LSTM {f} (layer {l})")
            if not t.is_contiguous():
                raise SumkError(f"decoder LSTM {f} (layer {l}) must be contiguous")
            setattr(arr[l], f, t.data_ptr())
    return arr


def lstm_decoder_forward(sb, layers, H, h0=None, c0=None):
    """Step-wise autoregressive decoder (SumGAN's dLSTM): layers = [(w_ih (4H,H), w_hh, b_ih, b_hh)] * L; h0 / c0 (L, n_seq, H)
    or None.  Returns (top-layer outputs (n_rows, H) in time order, workspace)."""
    lib = _lib.load()
    L = len(layers)
    dev = layers[0][0].device
    w = _dir_array(_lib.LstmDirWeights, layers)
    for name, t in (("h0", h0), ("c0", c0)):
        if t is not None:
            _require_gpu(t, f"decoder {name}")
            if tuple(t.shape) != (L, sb.n_seq, H) or not t.is_contiguous():
                raise SumkError(f"decoder {name} must be contiguous ({L}, {sb.n_seq}, {H}), got {tuple(t.shape)}")
    nbytes = lib.sumk_lstm_decoder_workspace_bytes(H, L, sb.n_seq, sb.off_host_p)
    if nbytes == 0:
        _lib.check(-1, "sumk_lstm_decoder_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty(sb.n_rows, H, dtype=torch.float32, device=dev)
    _lib.check(lib.sumk_lstm_decoder_forward(H, L, sb.n_seq, sb.off_host_p, sb.off_dev_p, w, _p(h0), _p(c0), _p(out), _p(ws), nbytes,
                                             _stream()), "sumk_lstm_decoder_forward")
    return out, ws


def lstm_decoder_backward(sb, layers, H, c0, out, dout, grads, ws, want_d0):
    """grads = [(dw_ih, dw_hh, db_ih, db_hh)] * L, ACCUMULATED.  Returns (dh0, dc0) (L, n_seq, H) or (None, None)."""
    lib = _lib.load()
    L = len(layers)
    w = _dir_array(_lib.LstmDirWeights, layers)
    g = _dir_array(_lib.LstmDirGrads, grads)
    dh0 = torch.empty(L, sb.n_seq, H, dtype=torch.float32, device=out.device) if want_d0 else None
    dc0 = torch.empty(L, sb.n_seq, H, dtype=torch.float32, device=out.device) if want_d0 else None
    _lib.check(lib.sumk_lstm_decoder_backward(H, L, sb.n_seq, sb.off_host_p, sb.off_dev_p, w, _p(c0), _p(out), _p(dout.contiguous()), g,
                                              _p(dh0), _p(dc0), _p(ws), ws.numel(), _stream()), "sumk_lstm_decoder_backward")
    return dh0, dc0


# ------------------------------------------------------------------------------------------------ GRU cell (DSN(cell="gru"))
def gru_cell_forward(gx, gh, h_prev, mask, save=True):
    """One fused GRU step for B sequences: gx, gh (B, 3H), h_prev (B, H), mask (B,) or None -> (h (B, H), rzn (B, 3H) or None)."""
    lib = _lib.load()
    _require_gpu(gx, "gru_cell gx")
    B, H = h_prev.shape
    h = torch.empty_like(h_prev)
    rzn = torch.empty_like(gx) if save else None
    _lib.check(lib.sumk_gru_cell_forward(_p(gx), _p(gh), _p(h_prev), _p(mask), _p(h), _p(rzn), B, H, _stream()), "sumk_gru_cell_forward")
    return h, rzn


def gru_cell_backward(dh, rzn, gh, h_prev, mask):
    """-> (dgx (B, 3H), dgh (B, 3H), direct part of dh_prev (B, H))."""
    lib = _lib.load()
    B, H = h_prev.shape
    dgx, dgh, dhp = torch.empty_like(gh), torch.empty_like(gh), torch.empty_like(h_prev)
    _lib.check(lib.sumk_gru_cell_backward(_p(dh.contiguous()), _p(rzn), _p(gh), _p(h_prev), _p(mask), _p(dgx), _p(dgh), _p(dhp), B, H,
                                          _stream()), "sumk_gru_cell_backward")
    return dgx, dgh, dhp
