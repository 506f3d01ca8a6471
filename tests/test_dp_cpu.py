"""CPU (gloo, world_size 2): the data-parallel plumbing -- video->rank sharding and the single flat-bucket gradient
all-reduce -- without any compute call (the HIP path needs a GPU)."""
import os
import socket
import numpy as np
import torch
import torch.multiprocessing as mp

from summarizer_amd.training import shard_keys, all_reduce_flat, dist_info


def test_shard_keys_partition_and_balance():
    rng = np.random.default_rng(0)
    lens = [int(v) for v in rng.integers(100, 700, 40)]
    keys = [f"video_{i+1}" for i in range(40)]
    for world in (1, 2, 4, 8):
        parts = [shard_keys(keys, lens, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == sorted(keys)                       # disjoint cover
        loads = [sum(lens[keys.index(k)] for k in p) for p in parts]
        assert max(loads) - min(loads) <= max(lens)                          # greedy longest-first bound
        assert parts == [shard_keys(keys, lens, r, world) for r in range(world)]   # deterministic


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert dist_info() == (rank, world)
        bucket = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        scale = all_reduce_flat(bucket)
        avg = bucket * scale
        expect = torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok = bool(torch.allclose(avg, expect))
        keys = [f"video_{i}" for i in range(11)]
        lens = [50 + 7 * i for i in range(11)]
        mine = shard_keys(keys, lens, rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ok = ok and sorted(sum(gathered, [])) == sorted(keys)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_flat_bucket_allreduce_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_dp_schedule_prediction_and_auto_batch_videos():
    """training.predicted_dp_efficiency / choose_batch_videos / resolve_batch_videos: a single process keeps the
    reference's one-video-per-step schedule; under torch.distributed the trainers say what a step is predicted to cost and default to
    `batch_videos=auto`: the smallest batch predicted >= 0.9 under the pessimistic (ring) all-reduce estimate, capped by the rank's shard."""
    from summarizer_amd import training as T
    for kind, prec in (("vasnet", "fp32"), ("vasnet", "bf16"), ("dsn", "fp32")):
        prev = 0.0
        for bv in (1, 2, 8, 32, 128):
            lo, hi = T.predicted_dp_efficiency(kind, prec, 8, bv)
            assert 0.0 < lo <= hi <= 1.0 and lo >= prev
            prev = lo
        bv = T.choose_batch_videos(kind, prec, 8, 1000)
        assert T.predicted_dp_efficiency(kind, prec, 8, bv)[0] >= 0.9
        assert bv == 1 or T.predicted_dp_efficiency(kind, prec, 8, bv - 1)[0] < 0.9
        assert T.choose_batch_videos(kind, prec, 1, 1000) == 1
        assert T.choose_batch_videos(kind, prec, 8, 3) <= 3                 # capped by the shard
    lo1, _ = T.predicted_dp_efficiency("vasnet", "fp32", 8, 1)
    assert lo1 < 0.9                                                        # the default schedule is NOT predicted to scale at 0.9: said, not hidden
    ring, direct = T.predicted_allreduce_us(21.0e6, 8)
    assert 200 < ring < 300 and 30 < direct < 80
    assert T.resolve_batch_videos({}, "vasnet", "fp32", list(range(40))) == 1
    assert T.resolve_batch_videos({"batch_videos": "4"}, "vasnet", "fp32", list(range(40))) == 4
    assert T.resolve_batch_videos({"batch_videos": "auto"}, "vasnet", "fp32", list(range(40))) == 1      # single process: nothing to hide
    # round 6: under torch.distributed the DEFAULT is "auto" (the one-video schedule stays reachable as batch_videos=1)
    real = T.dist_info
    T.dist_info = lambda: (0, 8)
    try:
        keys = list(range(400))
        bv = T.resolve_batch_videos({}, "vasnet", "fp32", keys)
        assert bv > 1 and T.predicted_dp_efficiency("vasnet", "fp32", 8, bv)[0] >= 0.9
        assert T.resolve_batch_videos({"batch_videos": "1"}, "vasnet", "fp32", keys) == 1
        assert T.resolve_batch_videos({}, "vasnet", "fp32", list(range(16))) <= 2                     # capped by the rank's shard (16 videos / 8 ranks)
    finally:
        T.dist_info = real
