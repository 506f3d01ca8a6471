"""Training-side plumbing shared by the trainers: a flat parameter/gradient bucket with a fused HIP Adam step, and the
data-parallel (one process per GPU, RCCL over xGMI) pieces -- video->rank sharding and ONE all-reduce of the flat
gradient bucket per optimiser step.  The reference has no distributed code; this is the MI355X-native
scale-out of its per-video loop: videos are the independent units, gradients are the only exchange."""
import math
import torch

from . import kernels


def dist_info():
    """(rank, world_size) of the current torch.distributed job, (0, 1) when not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_keys(keys, lens, rank, world):
    """Greedy longest-first assignment of videos to ranks balancing sum of frames; deterministic, every
    rank computes the same partition.  Returns the keys owned by `rank` (original relative order kept)."""
    if world == 1:
        return list(keys)
    order = sorted(range(len(keys)), key=lambda i: (-lens[i], i))
    load = [0] * world
    owner = [0] * len(keys)
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += lens[i]
    return [k for i, k in enumerate(keys) if owner[i] == rank]


def all_reduce_flat(bucket, average=True):
    """SUM all-reduce of one flat bucket over the default process group (RCCL on GPUs, gloo in CPU tests).
    Returns the factor the optimiser step still has to apply: 1/world_size when `average` (folded into the fused Adam
    kernel, not an extra pass over the bucket), 1.0 when the caller already normalised its loss by the GLOBAL number of
    videos of the step (`step_video_total`)."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        return 1.0 / world if average else 1.0
    return 1.0


def shard_sizes(keys, lens, world):
    """Number of videos every rank owns under `shard_keys` (each rank computes the same list, no communication)."""
    return [len(shard_keys(keys, lens, r, world)) for r in range(world)]


def plan_shards(train_keys, lens_fn, bv):
    """(my_keys, sizes, steps_per_epoch) of this rank for a fold: the static video -> rank assignment (DSN's per-video
    baselines therefore stay rank-local), every rank's shard size, and the number of optimiser steps per epoch -- padded to
    the largest shard so that all ranks enter the same number of collectives.  lens_fn() -> frames per key (read only when
    world > 1)."""
    import math
    rank, world = dist_info()
    if world == 1:
        return list(train_keys), [len(train_keys)], math.ceil(len(train_keys) / bv)
    lens = lens_fn()
    sizes = shard_sizes(train_keys, lens, world)
    return shard_keys(train_keys, lens, rank, world), sizes, max(1, math.ceil(max(sizes) / bv))


def step_video_total(sizes, bv, step):
    """Videos ALL ranks contribute at optimiser step `step` of an epoch when rank r walks its shard of sizes[r] videos in
    slices of `bv`.  Ranks whose shard has run out contribute none: dividing the summed gradient by this count (not by
    world * bv) keeps the ragged tail steps of an epoch correctly averaged and every video equally weighted."""
    return sum(max(0, min(bv, n - step * bv)) for n in sizes)


# ---- what a data-parallel step is expected to cost (NOT measured on more than one GPU yet) -------------------------
# Per-GPU step time of the trainers as a function of the videos per rank per step (TVSum-shaped videos, ~240 frames, D = 1024; fitted to the
# 1-GPU measurements of bench.py: 1 video and 50 videos per step), the bucket a step all-reduces, and the fraction of the exchange the step
# hides (VASNet: the tail piece, 40 % of the bucket, runs under the attention backward; DSN: the reverse direction's half runs under the
# forward direction's weight gradients).  All-reduce time: SURVEY section 5's link model -- a ring is per-link bound (2 (N - 1) / N bytes
# over ~153 GB/s), a direct reduce-scatter + all-gather moves bytes / N per link and phase.
_DP_MODEL = {            # (us per step at 1 video, extra us per further video, bucket bytes, hidden fraction)
    ("vasnet", "fp32"): (300.0, 55.0, 21.0e6, 0.4),
    ("vasnet", "bf16"): (260.0, 10.0, 10.5e6, 0.4),
    ("dsn", "fp32"): (1700.0, 40.0, 10.5e6, 0.5),
}


def predicted_allreduce_us(nbytes, world):
    """(ring, direct) estimate of one SUM all-reduce of nbytes over `world` GPUs of one xGMI node, microseconds."""
    link = 153.0e9
    ring = 2.0 * (world - 1) / world * nbytes / link * 1e6 + 15.0
    direct = 2.0 * nbytes / world / link * 1e6 + 15.0
    return ring, direct


def predicted_dp_efficiency(kind, precision, world, batch_videos):
    """[pessimistic (ring), optimistic (direct)] predicted weak-scaling efficiency step / (step + exposed exchange) of a trainer's step
    with `batch_videos` videos per rank.  A prediction to compare the first multi-GPU measurement with, nothing more."""
    t1, dt, nbytes, hidden = _DP_MODEL.get((kind, precision), _DP_MODEL[(kind, "fp32")])
    step = t1 + dt * (batch_videos - 1)
    return [round(step / (step + (1.0 - hidden) * ar), 3) for ar in predicted_allreduce_us(nbytes, world)]


def choose_batch_videos(kind, precision, world, shard_videos, target=0.9):
    """Smallest number of videos per rank per step whose PESSIMISTIC predicted efficiency reaches `target`, capped by the rank's shard
    (extra_params batch_videos=auto; the trainers' default under torch.distributed since round 6).  A single process keeps 1 -- the
    reference's optimisation schedule -- because the videos per step change the training dynamics, not only the speed."""
    if world <= 1:
        return 1
    for bv in range(1, max(1, shard_videos) + 1):
        if predicted_dp_efficiency(kind, precision, world, bv)[0] >= target:
            return bv
    return max(1, shard_videos)


def resolve_batch_videos(extra_params, kind, precision, train_keys, log=None):
    """batch_videos of a trainer run: extra_params["batch_videos"] = an integer or "auto" (choose_batch_videos).  DEFAULT: 1 in a single
    process -- the reference's schedule, one optimiser step per video (vasnet.py:193-212) -- and "auto" under torch.distributed (round 6):
    one video per rank per step is ~0.3 ms of compute against a 21 MB gradient exchange, predicted at 0.68-0.89 weak-scaling efficiency,
    and that would be the first number a multi-GPU run shows; `batch_videos=1` stays reachable explicitly.  Under torch.distributed the
    choice and the predicted efficiency of the step are logged, so that a slow multi-GPU run explains itself."""
    rank, world = dist_info()
    raw = str(extra_params.get("batch_videos", "auto" if world > 1 else 1))
    shard = max(1, -(-len(train_keys) // world))
    bv = choose_batch_videos(kind, precision, world, shard) if raw == "auto" else int(raw)
    if world > 1 and log is not None:
        lo, hi = predicted_dp_efficiency(kind, precision, world, bv)
        how = "given" if "batch_videos" in extra_params and raw != "auto" else "auto: the smallest value predicted >= 0.9 under the ring estimate, capped by the rank's shard"
        log.info(f"data parallel over {world} ranks: batch_videos={bv} per rank ({how}; global batch {bv * world} videos), predicted weak-scaling "
                 f"efficiency {lo}-{hi} (ring - direct all-reduce); extra_params batch_videos=1 keeps the reference's one-video steps")
    return bv


class RcclDirect:
    """The flat-bucket all-reduce through libsumk's own RCCL entry point (`sumk_allreduce_flat`) instead of torch.distributed:
    the collective is enqueued on the CURRENT HIP stream like any other kernel of the step (no process-group stream hop).
    The 128-byte RCCL id travels over the torch.distributed group that already exists (broadcast_object_list); one
    communicator per process.  Opt-in: SUMK_RCCL_DIRECT=1 (FlatAdam picks it up); torch.distributed stays the default."""
    _instance = None

    def __init__(self):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self._lib, self._C = _lib.load(), C
        rank, world = dist_info()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(self._lib.sumk_comm_unique_id(ident), "sumk_comm_unique_id")
        box = [bytes(ident)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        comm = C.c_void_p()
        _lib.check(self._lib.sumk_comm_init(ident, rank, world, C.byref(comm)), "sumk_comm_init")
        self.comm, self.world = comm, world

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = RcclDirect()
        return cls._instance

    def all_reduce(self, buf):
        """SUM all-reduce of a contiguous fp32 / bf16 device tensor, in place, on the current stream."""
        from . import _lib
        dtype = {torch.float32: 0, torch.bfloat16: 1}[buf.dtype]
        _lib.check(self._lib.sumk_allreduce_flat(self.comm, self._C.c_void_p(buf.data_ptr()), buf.numel(), dtype,
                                                 self._C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)), "sumk_allreduce_flat")

    def close(self):
        from . import _lib
        if self.comm:
            _lib.check(self._lib.sumk_comm_destroy(self.comm), "sumk_comm_destroy")
            self.comm = None
        RcclDirect._instance = None


def _collective_sum(t):
    """One SUM all-reduce of tensor `t`: torch.distributed, or libsumk's RCCL entry point under SUMK_RCCL_DIRECT=1 (GPU tensors)."""
    import os
    import torch.distributed as dist
    if t.is_cuda and os.environ.get("SUMK_RCCL_DIRECT") == "1":
        RcclDirect.get().all_reduce(t)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


class FlatAdam:
    """torch.optim.Adam(params, lr, weight_decay) semantics (vasnet.py:181, dsn.py:70-73) over ONE flat fp32 bucket:
    parameters and their .grad become views of two contiguous buffers, so
      * zero_grad() is one memset,
      * the data-parallel exchange is one all-reduce of `flat_grad` (RCCL; 21 MB for VASNet, 10.5 MB for DSN),
      * step() is one HIP kernel (sumk_adam_step) -- optionally with the clip_grad_norm_ scale folded in (dsn.py:145).
    The gradient tensors autograd produces are accumulated into the views in place."""

    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, comm_dtype=None):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        if not self.params[0].is_cuda:
            raise kernels.SumkError("FlatAdam needs GPU parameters (HIP Adam kernel; no CPU fallback)")
        n = sum(p.numel() for p in self.params)
        n_pad = (n + 3) // 4 * 4
        self.flat_param = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p)
                if p.grad is not None:             # like torch.optim: constructing an optimiser does not touch gradients
                    self.flat_grad[off:off + k].copy_(p.grad.reshape(-1))
                p.grad = self.flat_grad[off:off + k].view_as(p)
                off += k
        self.n = n
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.step_count = 0
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._state = torch.zeros(4, dtype=torch.int32, device=dev)     # [0]: steps taken, kept on the device (sumk_adam_step_dev)
        self._side, self._tail_from = None, None      # side stream / split point of an in-flight early all-reduce
        self._zero_by_step = False                    # the last thing that touched the gradient bucket was step(zero_grad=True)
        # mixed-precision mode: the gradient bucket crosses the all-reduce as bf16 (half the bytes over xGMI); the fp32 bucket,
        # the moments and the master weights stay fp32.  The bf16 staging buffer exists only under torch.distributed.
        if comm_dtype not in (None, torch.float32, torch.bfloat16):
            raise kernels.SumkError(f"FlatAdam: comm_dtype must be None, float32 or bfloat16, got {comm_dtype}")
        self.comm_dtype = torch.bfloat16 if comm_dtype == torch.bfloat16 else torch.float32
        self._comm = None

    def zero_grad(self, zeroed_by_step=False):
        """Zero the gradient bucket.  zeroed_by_step=True is the CALLER's statement that nothing has written a gradient since the last
        step(zero_grad=True) -- the step -> zero_grad -> backward order of the trainers -- in which case the Adam kernel already left the
        bucket zero and the fill launch is skipped (once; and never while a stream capture is recording: a captured fill must be IN the
        graph).  Without that statement the bucket is always filled: the optimiser cannot see a backward pass that ran in between."""
        skip = zeroed_by_step and self._zero_by_step and not torch.cuda.is_current_stream_capturing()
        self._zero_by_step = False
        if not skip:
            self.flat_grad.zero_()
        off = 0
        for p in self.params:                      # re-attach views in case something replaced .grad
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + k].view_as(p)
            off += k

    def all_reduce_grads(self, average=True):
        """Sum the gradient bucket over the data-parallel group: ONE collective per optimiser step (plus the early one of
        `reduce_tail_async` when that was used).  Returns the scale the optimiser step folds into the gradient
        (1/world if `average`, else 1.0: the trainers normalise their loss by the global video count instead)."""
        import torch.distributed as dist
        rank, world = dist_info()
        if world == 1:
            return 1.0
        if self._tail_from is None:
            self._reduce(self.flat_grad)
        else:                                      # the tail is already in flight on the side stream: reduce the head, then join
            self._reduce(self.flat_grad[:self._tail_from])
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._side)
            self._tail_from = None
        return 1.0 / world if average else 1.0

    def _reduce(self, piece):
        """SUM all-reduce of a slice of the gradient bucket, through a bf16 staging buffer when comm_dtype is bfloat16."""
        import torch.distributed as dist
        if self.comm_dtype == torch.float32:
            _collective_sum(piece)
            return
        if self._comm is None:
            self._comm = torch.empty(self.flat_grad.numel(), dtype=torch.bfloat16, device=self.flat_grad.device)
        off = (piece.data_ptr() - self.flat_grad.data_ptr()) // 4
        stage = self._comm[off:off + piece.numel()]
        kernels.cast_f32_bf16(piece, stage)
        _collective_sum(stage)
        kernels.cast_bf16_f32(stage, piece)

    def tail_offset(self, first_param):
        """Element offset in the flat bucket of `first_param` (a Parameter of this optimiser)."""
        off = 0
        for p in self.params:
            if p is first_param:
                return off
            off += p.numel()
        raise KeyError("parameter is not in this optimiser")

    def reduce_tail_async(self, tail_from, ready_event):
        """Overlap: all-reduce flat_grad[tail_from:] on a side stream as soon as `ready_event` (recorded by the HIP backward
        once those gradients are final) has fired, while the rest of the backward still runs on the compute stream.
        No-op outside torch.distributed.  `all_reduce_grads` later reduces the head and joins the side stream."""
        import torch.distributed as dist
        rank, world = dist_info()
        if world == 1 or not self.flat_grad.is_cuda:
            return
        import os
        if os.environ.get("SUMK_RCCL_DIRECT") == "1":
            return        # ONE communicator: its collectives stay on one stream, in one order (no early piece; all_reduce_grads reduces the whole bucket)
        if self._side is None:
            self._side = torch.cuda.Stream(self.flat_grad.device)
        # keep both pieces 16-byte aligned -- rounding UP: the (up to 3) elements in between belong to the parameter BEFORE the tail,
        # whose gradient is not final yet when ready_event fires; they travel with the head reduce after the whole backward
        tail_from = -(-tail_from // 4) * 4
        self._side.wait_event(ready_event)
        with torch.cuda.stream(self._side):
            self._reduce(self.flat_grad[tail_from:])
        self._tail_from = tail_from

    def broadcast(self, src=0):
        """Identical weights on every rank: ONE broadcast of the flat parameter bucket."""
        import torch.distributed as dist
        if dist_info()[1] > 1:
            dist.broadcast(self.flat_param, src=src)

    def grad_norm(self, grad_scale=1.0):
        """L2 norm of (grad_scale * gradient bucket) -- one HIP reduction + one scalar D2H."""
        self._norm.zero_()
        kernels.sumsq(self.flat_grad, out=self._norm)
        return math.sqrt(float(self._norm.item())) * grad_scale

    def step(self, grad_scale=1.0, max_norm=None, zero_grad=False):
        """grad_scale: multiplies the gradient first (1/world_size of a DP average).  max_norm: clip_grad_norm_
        semantics applied AFTER the all-reduce, on the averaged gradient (torch: coef = max_norm/(norm+1e-6), clamped to 1).
        The step counter and the clip coefficient stay on the device (sumk_adam_step_dev): no host synchronisation, so a whole
        training step enqueues without waiting and can be captured into a HIP graph (the replays advance the counter)."""
        sumsq = None
        if max_norm is not None:
            self._norm.zero_()
            kernels.sumsq(self.flat_grad, out=self._norm)
            sumsq = self._norm
        self.step_count += 1          # host mirror (not advanced by graph replays; the device counter in _state[0] is authoritative)
        # zero_grad: the gradient bucket is left zero by the Adam kernel (the next step's zero_grad(), folded into this pass)
        kernels.adam_step_dev(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self._state, self.lr, self.betas,
                              self.eps, self.weight_decay, grad_scale, sumsq, 0.0 if max_norm is None else max_norm, zero_grad=zero_grad)
        self._zero_by_step = bool(zero_grad) and not torch.cuda.is_current_stream_capturing()      # (short-circuits: no driver query on plain steps)


def broadcast_parameters(model, src=0):
    """Identical initial weights on every rank for a model that has no FlatAdam yet: the parameters travel as
    ONE flat buffer (one collective), not one broadcast per tensor.  The trainers use `FlatAdam.broadcast()` instead."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world > 1:
        params = [p for p in model.parameters()]
        flat = torch.cat([p.data.reshape(-1) for p in params])
        dist.broadcast(flat, src=src)
        off = 0
        for p in params:
            p.data.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
