"""GPU, 2 processes on ONE device (gloo carries the collective; RCCL needs one GPU per rank, which the 1-GPU test box does
not have): the data-parallel training path end to end -- video sharding, parameter broadcast, ONE flat-bucket gradient
all-reduce per step folded into the fused Adam -- must give every rank the same weights, equal to a single process
stepping on the same two videos as one batch (mean of per-video losses == average of per-rank gradients)."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _run(rank, world, port, q, bv, precision="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import random
    import torch.distributed as dist
    from summarizer_amd.models.vasnet import VASNetTrainer
    from summarizer_amd.utils.datasets import synthetic_dataset
    from summarizer_amd.utils.hps import make_hps
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = synthetic_dataset(3, seed=9, D=128, t_range=(40, 90), n_users=4)
        keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))
        hps = make_hps(ds, [{"train_keys": keys[:2], "test_keys": keys[2:]}], epochs=2, test_every_epochs=5, lr=1e-3,
                       selection_algorithm="rank", extra_params={"input_size": "128", "batch_videos": str(bv), "precision": precision})
        torch.manual_seed(100 + rank)          # DIFFERENT init per rank: broadcast_parameters must make them agree
        random.seed(5)
        tr = VASNetTrainer(hps, hps.splits_files[0]).reset()
        tr.model.dropout.p = 0.0
        if world == 1:
            torch.manual_seed(100); tr = VASNetTrainer(hps, hps.splits_files[0]).reset(); tr.model.dropout.p = 0.0
        tr.train(0)
        q.put((rank, {k: v.detach().cpu().numpy() for k, v in tr.model.state_dict().items()}))
    finally:
        if world > 1:
            dist.destroy_process_group()


def _spawn(world, bv, precision="fp32"):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, world, port, q, bv, precision)) for r in range(world)]
    for p in procs: p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs: p.join(timeout=120)
    return res


def test_dp_two_ranks_equal_single_process_batch_of_two():
    dp = _spawn(2, 1)
    single = _spawn(1, 2)[0]
    for k in dp[0]:
        np.testing.assert_array_equal(dp[0][k], dp[1][k], err_msg=f"ranks disagree on {k}")
        np.testing.assert_allclose(dp[0][k], single[k], atol=2e-6, err_msg=f"DP != single-process batch for {k}")


def test_dp_mixed_precision_bf16_gradient_bucket():
    """precision "bf16": the gradient bucket crosses the all-reduce as bf16 (cast kernels + one bf16 collective, and the early
    tail piece on the side stream).  Both ranks must end bit-identical; against the single-process batch of two the weights agree
    to bf16-rounding of the gradients (Adam steps of lr = 1e-3 => a few 1e-4 after 4 steps)."""
    dp = _spawn(2, 1, "bf16")
    single = _spawn(1, 2, "bf16")[0]
    for k in dp[0]:
        np.testing.assert_array_equal(dp[0][k], dp[1][k], err_msg=f"ranks disagree on {k}")
        np.testing.assert_allclose(dp[0][k], single[k], atol=2e-3, err_msg=f"DP(bf16 comm) far from single-process batch for {k}")


def test_bench_multi_rank_control_flow_on_one_gpu():
    """`bench.py` as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU) -- here with both ranks on the
    one GPU of the test box over gloo (SUMK_BENCH_ONE_GPU=1): barriers, max-over-ranks timing, ONE JSON line from rank 0 whose
    value counts the frames of all ranks."""
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SUMK_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--videos", "6"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak" and "cpu_baseline" not in out
    per_rank = out["config"]["frames_per_step_per_gpu"]
    assert abs(out["value"] - 2 * per_rank * 5 / (out["ms_per_step"] * 5 / 1e3)) / out["value"] < 1e-3
    assert out["roofline"]["frac"] > 0 and out["bf16x3_mode"]["max_abs_score_diff_vs_fp32"] < 1e-4
