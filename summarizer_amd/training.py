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


def all_reduce_flat(bucket):
    """SUM all-reduce of one flat bucket over the default process group (RCCL on GPUs, gloo in CPU tests).
    Returns 1/world_size -- the averaging factor is applied later inside the fused Adam kernel, not as an extra pass."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        return 1.0 / world
    return 1.0


class FlatAdam:
    """torch.optim.Adam(params, lr, weight_decay) semantics (vasnet.py:181, dsn.py:70-73) over ONE flat fp32 bucket:
    parameters and their .grad become views of two contiguous buffers, so
      * zero_grad() is one memset,
      * the data-parallel exchange is one all-reduce of `flat_grad` (RCCL; 21 MB for VASNet, 10.5 MB for DSN),
      * step() is one HIP kernel (sumk_adam_step) -- optionally with the clip_grad_norm_ scale folded in (dsn.py:145).
    The gradient tensors autograd produces are accumulated into the views in place."""

    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
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
                p.grad = self.flat_grad[off:off + k].view_as(p)
                off += k
        self.n = n
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.step_count = 0
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def zero_grad(self):
        self.flat_grad.zero_()
        off = 0
        for p in self.params:                      # re-attach views in case something replaced .grad
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + k].view_as(p)
            off += k

    def all_reduce_grads(self):
        """Average the gradient bucket over the data-parallel group: ONE collective per optimiser step.  Returns the
        scale (1/world) the optimiser step folds into the gradient."""
        return all_reduce_flat(self.flat_grad)

    def grad_norm(self, grad_scale=1.0):
        """L2 norm of (grad_scale * gradient bucket) -- one HIP reduction + one scalar D2H."""
        self._norm.zero_()
        kernels.sumsq(self.flat_grad, out=self._norm)
        return math.sqrt(float(self._norm.item())) * grad_scale

    def step(self, grad_scale=1.0, max_norm=None):
        """grad_scale: multiplies the gradient first (1/world_size of a DP average).  max_norm: clip_grad_norm_
        semantics applied AFTER the all-reduce, on the averaged gradient (torch: coef = max_norm/(norm+1e-6), clamped to 1)."""
        if max_norm is not None:
            norm = self.grad_norm(grad_scale)
            coef = min(1.0, max_norm / (norm + 1e-6))
            grad_scale = grad_scale * coef
        self.step_count += 1
        kernels.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                          self.betas, self.eps, self.weight_decay, grad_scale)


def broadcast_parameters(model, src=0):
    """Identical initial weights on every rank."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=src)
