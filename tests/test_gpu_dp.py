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


def _run_dsn(rank, world, port, q, bv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import random
    import zlib
    import torch.distributed as dist
    from summarizer_amd.models.dsn import DSNTrainer
    from summarizer_amd.utils.datasets import synthetic_dataset
    from summarizer_amd.utils.hps import make_hps
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = synthetic_dataset(3, seed=11, D=128, t_range=(40, 90), n_users=4)
        keys = sorted(ds.keys(), key=lambda k: int(k.split("_")[1]))

        class ReplayDSNTrainer(DSNTrainer):
            """Bernoulli draws as a pure function of (video key, how often that video has been drawn): the same episodes
            whichever rank / batch composition processes the video (the device generators of two processes cannot be matched)."""
            def _sample_actions(self, dist_, n_episodes, keys_):
                self._draws = getattr(self, "_draws", {})
                out, off = [], 0
                for k in keys_:
                    T = self.dataset[k]["features"].shape[0]
                    c = self._draws[k] = self._draws.get(k, 0) + 1
                    u = np.random.default_rng([zlib.crc32(k.encode()), c]).random((n_episodes, T)).astype(np.float32)
                    out.append((torch.from_numpy(u).to(dist_.probs.device) < dist_.probs[off:off + T]).float())
                    off += T
                return torch.cat(out, dim=1)

        hps = make_hps(ds, [{"train_keys": keys[:2], "test_keys": keys[2:]}], epochs=3, test_every_epochs=5, lr=1e-3,
                       selection_algorithm="rank",
                       extra_params={"input_size": "128", "hidden_size": "24", "num_episodes": "3", "sup": True, "batch_videos": str(bv)})
        torch.manual_seed(200 + rank); random.seed(5)      # DIFFERENT init per rank: the flat broadcast must make them agree
        tr = ReplayDSNTrainer(hps, hps.splits_files[0]).reset()
        if world == 1:
            torch.manual_seed(200); tr = ReplayDSNTrainer(hps, hps.splits_files[0]).reset()
        tr.train(0)
        rewards = [v for _, v in hps.writer.scalars["synthetic/Fold_1/Train/Reward"]]
        q.put((rank, {k: v.detach().cpu().numpy() for k, v in tr.model.state_dict().items()}, rewards))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_dsn_reinforce_dp_two_ranks_equal_single_process_batch_of_two():
    """DSNTrainer under data parallelism (BASELINE config 4 in miniature): per-video baselines stay on the rank that owns the
    video, the flat bucket is all-reduced once per step and clip_grad_norm_(5.0) is applied AFTER the reduction, on the
    globally averaged gradient -- so two ranks with one video each must follow a single process stepping on both videos."""
    def spawn(world, bv):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_run_dsn, args=(r, world, port, q, bv)) for r in range(world)]
        for p in procs: p.start()
        res = {r: (w, rew) for r, w, rew in (q.get(timeout=300) for _ in procs)}
        for p in procs: p.join(timeout=120)
        return res
    dp = spawn(2, 1)
    single = spawn(1, 2)[0]
    for k in dp[0][0]:
        np.testing.assert_array_equal(dp[0][0][k], dp[1][0][k], err_msg=f"ranks disagree on {k}")
        np.testing.assert_allclose(dp[0][0][k], single[0][k], atol=5e-6, err_msg=f"DP != single-process batch for {k}")
    # each rank logs the reward of ITS video; the single process logs the mean over both
    np.testing.assert_allclose(np.mean([dp[0][1], dp[1][1]], axis=0), single[1], atol=1e-5)


def test_bench_train_mode_two_ranks_on_one_gpu():
    """`bench.py --mode train --gpus 2` as the driver would launch it (gloo, both ranks on the test box's one GPU): every step
    ends in the flat-bucket gradient all-reduce; one JSON line whose value counts both ranks' frames."""
    import json, subprocess, sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SUMK_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--videos", "6",
           "--mode", "train"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and "train" in out["config"]["workload"]
    assert abs(out["value"] - 2 * out["config"]["frames_per_step_per_gpu"] * 4 / (out["ms_per_step"] * 4 / 1e3)) / out["value"] < 1e-3


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
    # the scaling line also carries a data-parallel TRAINING leg, so that a multi-GPU run measures the gradient all-reduce
    tl = out["train_step_mode"]
    assert tl["frames_per_s"] > 0 and tl["allreduce_bytes_per_step"] > 20e6 and tl["collectives_per_step"] == 2      # tail piece early, head after the backward
    # ... and what a SCALE run needs to decompose it: the exchange alone, the step without it, exposed and hidden communication
    for leg in (tl, out["train_step_bf16_mode"], out["dsn_reinforce_step_mode"]):
        assert leg["allreduce_alone_us"] > 0 and leg["ms_per_step_without_allreduce"] > 0
        assert leg["exposed_comm_us"] >= 0 and leg["overlap_hidden_us"] >= 0
    tb = out["train_step_bf16_mode"]              # BASELINE config 2: bf16 products and a bf16 gradient bucket (half the bytes)
    assert tb["frames_per_s"] > 0 and 10e6 < tb["allreduce_bytes_per_step"] < 11e6
    rl = out["dsn_reinforce_step_mode"]           # BASELINE config 4: DSN REINFORCE data-parallel (10.5 MB bucket, clip after the reduce)
    assert "error" not in rl, rl
    # (round 5: the bucket's tail [reverse direction | head] goes out early on a side stream, the rest after the backward: two collectives)
    assert rl["frames_per_s"] > 0 and 10e6 < rl["allreduce_bytes_per_step"] < 11e6 and rl["collectives_per_step"] == 2
    ov = out["dp_one_video_per_rank_mode"]        # the trainers' DEFAULT data-parallel schedule: one video per rank per step, the 21 MB bucket every step
    assert "error" not in ov, ov
    assert ov["videos_per_rank_per_step"] == 1 and ov["frames_per_s"] > 0 and ov["allreduce_bytes_per_step"] > 20e6 and ov["collectives_per_step"] == 2
    assert ov["allreduce_alone_us"] > 0 and ov["exposed_comm_us"] >= 0 and 0 < ov["predicted_efficiency_at_this_world"][0] <= ov["predicted_efficiency_at_this_world"][1] <= 1
    assert out["ranks_seen"] == 2 and out["collective_backend"] == "gloo"


def test_bench_plain_python_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (VERDICT r2 weak #9a): the parent must start two rank processes itself -- before
    it touches the GPU -- and relay rank 0's single JSON line; n_gpus must be 2, not 1."""
    import json, subprocess, sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SUMK_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--videos", "6",
                        "--headline-only"], env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 4
    per_rank = out["config"]["frames_per_step_per_gpu"]
    assert abs(out["value"] - 2 * per_rank * 4 / (out["ms_per_step"] * 4 / 1e3)) / out["value"] < 1e-3
