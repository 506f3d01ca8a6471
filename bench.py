#!/usr/bin/env python3
"""bench.py -- frames scored / second for the VASNet scoring path on MI355X (BASELINE.json metric).

One "step" = one pass of the scorer over one packed batch of synthetic TVSum-shaped videos (S-TVSum):
50 videos, T_i = ceil(U(150,320)) from numpy default_rng(0), D = 1024 fp32 features already resident in HBM.
N > 1: one process per GPU (torchrun), each rank scores its OWN 50 videos (sharded by video, no data-path
collective) -> "scaling": "weak"; value = frames all ranks scored / max-over-ranks time.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
HBM_PEAK_GBS = 8000.0
# Untimed calls in front of a VASNet side leg's timed steps.  Five (rounds 1-6) left the first split-bf16 leg behind the fp32 headline region with ~40 us per
# step of clock / cache warm-up inside a 20-step timed region (bf16x6 0.724 ms against 0.683 with forty, same box, same flags: profiles/r06c_leg_warmup_probe.txt);
# the headline's own warm-up is the contract's W and is not touched.
LEG_WARMUP = int(os.environ.get("SUMK_BENCH_LEG_WARMUP", "40"))


def _k_one(loss):
    from summarizer_amd import kernels as _kk
    return _kk.one(loss.device)


_VERBOSE_KEYS = ("note", "traffic_source", "pmc_reference", "flops_per_launch", "mfma_flops_per_launch", "launches", "sample_detail")


def _compact(o, top=True):
    """The JSON line without free text: every leg's numbers stay, notes / provenance go (--notes keeps them).  The contract's own text
    fields (metric, unit, config.workload, cpu_baseline.sample, roofline.kernel) stay."""
    if isinstance(o, dict):
        # (the legs' own cpu_baseline objects keep value / cores / kind; their `sample` text stays for the headline's only)
        return {k: _compact(v, k == "cpu_baseline" and top) for k, v in o.items() if k not in _VERBOSE_KEYS and not (k == "sample" and not top and o.get("kind") and "value" in o)}
    if isinstance(o, list):
        return [_compact(v, False) for v in o]
    if isinstance(o, float) and o != 0.0 and abs(o) < 1e-3:
        return float(f"{o:.3g}")          # (score differences: three significant digits)
    return o


def tvsum_lens(n_videos=50):
    return [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, n_videos)]


def cpu_baseline(lens, D, budget_s=20.0, kind="vasnet", gpu_scores=None, seed_base=0, threads=None):
    """Reference-equivalent stock-PyTorch CPU path (oracle/torch_port.py), one video per call as in
    Trainer.test (summarizer/models/__init__.py:45-54).  torch's intra-op pool is swept over a few thread
    counts (a 256-thread pool is far slower than 16-32 threads on (T<=320, 1024) matrices); the BEST setting is
    reported as `value`, with `cores` = the threads that setting used.  Bounded sample (~budget_s seconds)."""
    import recipes as R
    from oracle import torch_port
    torch.manual_seed(1234)
    if kind == "vasnet":
        from summarizer_amd.models.vasnet import VASNet
        p = {k: v.detach() for k, v in VASNet(input_size=D).named_parameters()}
        score = lambda x: torch_port.vasnet_scores(x, p)
    else:                                   # DSN (BiLSTM 1024 -> 2 x 256) or sLSTM (2 layers, 2 x 1024): torch's own nn.LSTM on the CPU
        from summarizer_amd.models.dsn import DSN
        from summarizer_amd.models.sumgan import sLSTM
        m = DSN(input_size=D) if kind == "dsn" else sLSTM(input_size=D)
        p = {k: v.detach() for k, v in m.named_parameters()}
        pre, hw, hb = ("rnn.", "out.0.weight", "out.0.bias") if kind == "dsn" else ("lstm.", "out.weight", "out.bias")
        lstm = torch_port.make_lstm(p, pre, D, m.hidden_size, m.num_layers)
        score = lambda x: torch_port.bilstm_scores(x, p, pre, hw, hb, D, m.hidden_size, m.num_layers, lstm=lstm)
    ncores = os.cpu_count() or 1
    xs = [torch.from_numpy(R.features(T, 1, D, seed_base + i)) for i, T in enumerate(lens)]      # = the GPU batch of that rank (main: 1000 * rank + i)
    # (an intra-op pool as wide as a 256-cpu host is pathological on these sizes -- 95 frames/s for VASNet, minutes per video for
    #  the LSTMs -- so the sweep stops at 64 threads)
    cands = sorted({t for t in (threads or (1, 8, 16, 32, 64, min(ncores, 64))) if t <= ncores}) or [1]
    per = budget_s / len(cands)
    res = {}
    with torch.no_grad():
        for nt in cands:
            torch.set_num_threads(nt)
            score(xs[0])
            frames, t0, n = 0, time.perf_counter(), 0
            while True:
                x = xs[n % len(xs)]
                score(x)
                frames += x.shape[0]; n += 1
                el = time.perf_counter() - t0
                if el > per:
                    break
            res[nt] = (frames / el, n, el)
    best = max(res, key=lambda k: res[k][0])
    parity = None
    if gpu_scores is not None:
        # the SAME weights (seed 1234 default constructor) and the SAME inputs as the timed GPU batch: every video of the headline
        # batch through the port, compared with what the HIP path just produced (north_star gate: 1e-4)
        torch.set_num_threads(best)
        with torch.no_grad():
            ref = torch.cat([score(x).reshape(-1) for x in xs])
        parity = float((gpu_scores.detach().cpu().reshape(-1) - ref).abs().max())
    return dict(value=round(res[best][0], 1), unit="frames/s", cores=best, kind="port", parity_max_abs_diff_vs_port=parity,
                sample=f"{sum(v[1] for v in res.values())} single-video {kind} forwards (S-TVSum lengths, D={D}, fp32, torch CPU ops) in "
                       f"{sum(v[2] for v in res.values()):.0f} s on a {ncores}-cpu host; best of intra-op threads {list(res)}",
                sample_detail="frames/s by intra-op threads: " + ", ".join(f"{k}t={v[0]:.0f} ({v[1]} videos/{v[2]:.1f}s)" for k, v in res.items()))


def alt_precision_leg(model, x, lens, ref_scores, steps, frames, precision="bf16x3"):
    """Opt-in split-bf16 arithmetic on the same batch, reported NEXT TO the fp32 headline (never as `value`): this rank's
    frames/s, the largest score difference from the fp32 path (gate: 1e-4) and a roofline record of ITS dominant kernel -- the
    plane-aware wide GEMM of the QKV projection (csrc/gemm_pw.hip), timed with the library's HIP event pairs in a separate short pass."""
    from summarizer_amd import _lib
    lib = _lib.load()
    model.precision = precision
    try:
        with torch.no_grad():
            for _ in range(LEG_WARMUP):
                s = model.score_packed(x, lens)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                s = model.score_packed(x, lens)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            lib.sumk_prof_read(_lib.PROF_GEMM_QKV, None, None, 1)
            lib.sumk_prof_enable(1 << _lib.PROF_GEMM_QKV)
            for _ in range(10):
                model.score_packed(x, lens)
            torch.cuda.synchronize()
            lib.sumk_prof_enable(0)
            ms = C.c_double(0); n = C.c_int64(0)
            lib.sumk_prof_read(_lib.PROF_GEMM_QKV, C.byref(ms), C.byref(n), 1)
    finally:
        model.precision = "fp32"
    D = x.shape[1]
    # what the timed steps do NOT contain: the x -> planes split (once per data set: the planes are kept with the feature tensor) and the
    # weight-plane build (once per weight change); a streaming caller pays the split on every batch
    from summarizer_amd import kernels as _kk
    npl = _kk.PLANES_OF[precision]
    for _ in range(3):
        _kk.split_planes(x, npl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        _kk.split_planes(x, npl)
    torch.cuda.synchronize()
    split_us = (time.perf_counter() - t0) / 10 * 1e6
    rec = dict(frames_per_s_this_gpu=round(frames / dt, 1), ms_per_step=round(dt * 1e3, 4),
               max_abs_score_diff_vs_fp32=float((s - ref_scores).abs().max()), split_planes_us=round(split_us, 1),
               resident_planes_bytes_per_frame=int(D * 2 * npl),
               note=("products as bf16 hi/lo splits (3 bf16 MFMAs), fp32 accumulate; select with --precision bf16x3" if precision == "bf16x3"
                     else "operands split EXACTLY into 3 bf16 planes, 6 bf16 MFMAs per product, fp32 accumulate: fp32-grade "
                          "(same error bound vs float64 as the fp32 MFMA path); select with --precision bf16x6") +
                    ".  " + f"{LEG_WARMUP} untimed calls precede the timed steps.  " + "x planes and weight planes built once in warm-up (per data set / per weight change), not in the timed steps: a caller that "
                    "streams new features pays split_planes_us per batch, and resident planes cost resident_planes_bytes_per_frame of HBM beside the fp32 features")
    if n.value > 0:
        us = ms.value / n.value * 1e3
        fl = 2.0 * frames * 3 * D * D
        mf = {"bf16x3": 3.0, "bf16x6": 6.0}[precision]
        peak = BF16_MFMA_PEAK_TFLOPS / mf
        rec["roofline"] = dict(bound="mfma", kernel=("gemm_pw16_kernel<2 planes, 192x256, 16x16x32 MFMA>" if precision == "bf16x3" else "gemm_pw_kernel<3 planes, 192x256, 32x32x16 MFMA>") + " (QKV projection on operand planes, planes out)",
                               achieved=round(fl / us / 1e6, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(fl / us / 1e6 / peak, 4),
                               avg_launch_us=round(us, 2), launches=int(n.value), flops_per_launch=fl, mfma_flops_per_launch=fl * mf,
                               traffic=None,
                               note=f"algorithmic fp32 FLOP of the product; the kernel issues {int(mf)} dense bf16 MFMA FLOP per algorithmic FLOP, so the "
                                    f"ceiling is the 2.5 PFLOP/s bf16 peak / {int(mf)}.  The chip holds ~1.7 GHz (not 2.4) under this load (in-kernel "
                                    "s_memtime / s_memrealtime stamps, profiles/r05_pw_stamps.txt): PMC passes under profiles/r06_pmc_pw_*")
        sus = mfma_sustained()
        if sus:
            rec["roofline"]["frac_of_sustained"] = round(fl / us / 1e6 / (sus["bf16_16x16x32" if precision == "bf16x3" else "bf16_32x32x16"] / mf), 4)
        # `traffic` as the headline's: from separate rocprofv3 --pmc passes of THIS launch (scripts/pmc_pass.sh + pmc_to_json.py), kept under profiles/
        tp = os.path.join(ROOT, "profiles", f"r06_pmc_pw_{'x3' if precision == 'bf16x3' else 'x6'}_qkv.json")
        if os.path.exists(tp) and frames == 12003 and D == 1024:
            pmc = json.load(open(tp))
            rec["roofline"]["traffic"] = pmc.get("gemm_qkv_hbm_bytes_per_launch")
            rec["roofline"]["traffic_source"] = f"static: profiles/{os.path.basename(tp)} ({pmc.get('kernel', '')[:70]}; algorithmic {pmc.get('algorithmic_bytes_per_launch')} B)"
    return rec


_SUSTAINED = None


def mfma_sustained():
    """The matrix-pipe rate THIS box sustains with nothing but MFMAs in flight (csrc/mfma_probe.hip; ~3 ms per kind, outside every timed region): the
    practical ceiling of an MFMA-bound launch here.  `peak` in the roofline objects stays the guide's figure; `frac_of_sustained` is against this one."""
    global _SUSTAINED
    if _SUSTAINED is None:
        from summarizer_amd import kernels as _k
        try:
            kinds = (("bf16_32x32x16", "bf16"), ("bf16_16x16x32", "bf16_16"), ("f32_32x32x2", "f32"))
            r = {name: round(_k.mfma_sustained_rate(kind, 8000, True)[0], 1) for name, kind in kinds}
            r["constant_operands"] = {name: round(_k.mfma_sustained_rate(kind, 8000, False)[0], 1) for name, kind in kinds}
            r.update(unit="TFLOP/s", note="measured live: one 512-thread workgroup per CU issuing only MFMAs (4 independent accumulators per wave, no memory traffic), "
                                          "8 000 x 16 MFMAs per wave, every MFMA of an iteration on its own pseudo-random operand registers (`constant_operands`: one constant "
                                          "pair feeding all of them -- the chip clocks higher when the operand buses do not toggle); the guide's dense peaks are 2500 (bf16) "
                                          "and 157.3 (f32) TFLOP/s at 2.4 GHz")
            _SUSTAINED = r
        except Exception as e:          # noqa: BLE001
            _SUSTAINED = dict(error=f"{type(e).__name__}: {e}"[:200])
    return _SUSTAINED if "error" not in _SUSTAINED else None


def folded_leg(model, x, lens, ref_scores, steps, frames, precision="fp32"):
    """Opt-in inference mode VASNet(fold_vo=True), exact fp32 MFMA, reported NEXT TO the headline (never as `value`): the value and
    output projections are folded into one matrix once per weight change, so the out-projection GEMM is not executed at all --
    the frames/s below is real, but it is bought with 18 % fewer executed FLOPs, not with a faster kernel."""
    model.fold_vo = True
    model.precision = precision
    try:
        with torch.no_grad():
            for _ in range(LEG_WARMUP):
                s = model.score_packed(x, lens)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                s = model.score_packed(x, lens)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
    finally:
        model.fold_vo = False
        model.precision = "fp32"
    return dict(frames_per_s_this_gpu=round(frames / dt, 1), ms_per_step=round(dt * 1e3, 4), precision=precision,
                max_abs_score_diff_vs_default=float((s - ref_scores).abs().max()),
                executed_flops_vs_default=round((8.0 * 1024 * 1024 + 4 * 250 * 1024) / (10.0 * 1024 * 1024 + 4 * 250 * 1024), 3),
                note="opt-in VASNet(fold_vo=True): Wvo = Wo.Wv folded once per weight change, out-projection GEMM not executed; "
                     "scores equal up to fp32 re-association")


def bench_sumgan(args, dev, rank, world, dist):
    """One step = the three updates of ONE video (T = 300, D = 1024) through SumGANTrainer.train_video at the reference's
    default sizes.  Non-headline: reported as frames of the video per second."""
    assert args.mode == "train", "--model sumgan measures the training step (scoring is --model slstm)"
    import recipes as R
    from summarizer_amd.models.sumgan import SumGANTrainer
    from summarizer_amd.utils.datasets import DictDataset
    from summarizer_amd.utils.hps import make_hps
    T, D = 300, 1024
    hps = make_hps(DictDataset({}), [{"train_keys": [], "test_keys": []}], epochs=1, extra_params={})
    torch.manual_seed(1234)
    tr = SumGANTrainer(hps, hps.splits_files[0]).reset()
    tr.model.train()
    tr.setup_optimizers()
    x = torch.from_numpy(R.features(T, 1, D, 1000 * rank)).to(dev)
    y = torch.rand(T, 1, 1, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        vals = tr.train_video(x, y, noisy=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vals = tr.train_video(x, y, noisy=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(bool(torch.isfinite(v).all()) for v in vals)
    if rank == 0:
        n_params = sum(p.numel() for p in tr.model.parameters())
        print(json.dumps(dict(metric="frames scored/sec (T x 1024)", value=round(T * world * args.steps / elapsed, 1), unit="frames/s",
                              n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3),
                              higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                              config=dict(workload=f"sumgan train: one video (T={T}, D={D}) per step = selector+encoder, decoder and "
                                                   f"discriminator updates, reference default sizes ({n_params/1e6:.0f} M parameters)",
                                          frames_per_step_per_gpu=T, parallelism=f"replicas x{world}"),
                              roofline=None, note="non-headline mode (SURVEY section 8f rank 4)")), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def make_reinforce_step(model, x, lens, dev):
    """One DSNTrainer step on the packed batch (dsn.py:96-156): probabilities, 5 Bernoulli episodes, reward kernel,
    policy-gradient loss, backward, [gradient all-reduce under data parallelism,] grad-norm clip folded into the flat Adam
    (applied AFTER the reduction), moving-average baselines.  BASELINE config 4 per GPU."""
    from torch.distributions import Bernoulli
    from summarizer_amd import kernels
    from summarizer_amd.training import FlatAdam
    from summarizer_amd.autograd import PolicyLossFunction
    from summarizer_amd.training import dist_info
    opt = FlatAdam(model.parameters(), lr=5e-5, weight_decay=1e-5)
    sb = kernels.SeqBatch.get(lens, dev)
    base = torch.zeros(len(lens), device=dev)
    # data parallel: the bucket's tail [reverse direction | head] goes out on a side stream under the forward direction's weight-gradient
    # GEMMs, as DSNTrainer issues it (sumk_lstm_layer_grads::tail_ready_event)
    tail_from = None
    if dist_info()[1] > 1:
        tail_from = opt.tail_offset(dict(model.named_parameters())["rnn.weight_ih_l0_reverse"])
        model.tail_grads_ready_event = torch.cuda.Event()

    def run_step(reduce=True):
        opt.zero_grad(zeroed_by_step=True)
        probs = model.score_packed(x, lens)
        dist_ = Bernoulli(probs, validate_args=False)          # (validation is a D2H sync per step)
        actions = dist_.sample((5,))
        rewards = kernels.dsn_reward(x, sb, actions.contiguous())
        loss = PolicyLossFunction.apply(probs, sb, actions, rewards, base, 0.01, 0.5).mean()    # dsn.py:113-140 in two HIP kernels
        loss.backward(gradient=_k_one(loss))
        if reduce and tail_from is not None:
            opt.reduce_tail_async(tail_from, model.tail_grads_ready_event)
        opt.step(grad_scale=opt.all_reduce_grads() if reduce else 1.0 / dist_info()[1], max_norm=5.0, zero_grad=True)
        base.mul_(0.9).add_(rewards.mean(dim=0), alpha=0.1)        # in place: also valid under --graph replays
        return loss.detach()
    return run_step, opt


def single_video_leg(dev, D=1024, T=300, iters=400):
    """The reference's OWN calling pattern, driver-timed: ONE TVSum-sized video per forward (models/__init__.py:45-54) and ONE
    optimiser step per video (vasnet.py:193-212, dsn.py:96-156) -- what `main.py` unchanged runs with the default batch_videos = 1.
    VASNet and DSN, scoring and training step, each eager (the Python call per video) and as a HIP-graph replay (what the trainers do
    from their second epoch on: the step of a video captured once, replayed).  us per video and frames/s."""
    import recipes as R
    from summarizer_amd import kernels
    from summarizer_amd.autograd import SegmentMseMeanFunction
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.models.vasnet import VASNet
    from summarizer_amd.training import FlatAdam
    x3 = torch.from_numpy(R.features(T, 1, D, 4242)).to(dev)            # (T, 1, D): the reference's input layout
    x2 = x3.view(T, D)
    target = torch.rand(T, device=dev)
    sb = kernels.SeqBatch.get([T], dev)

    def timed(fn, n=iters, warm=200):     # (a ~100 us call: 20 warm-up calls were 2 ms -- the clock had not settled when timing began)
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def graphed(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g.replay

    out = {}
    for name, ctor in (("vasnet", VASNet), ("dsn", DSN)):
        torch.manual_seed(1234)
        m = ctor(input_size=D).to(dev).eval()
        def score():
            with torch.no_grad():
                return m(x3)
        t_eager = timed(score)          # (a captured-graph replay of the scoring call was measured SLOWER than the eager call, 98.7 vs 93.8 us
        m.train()                       #  in round 4 -- nine short launches are not host-bound -- and is no longer reported)
        opt = FlatAdam(m.parameters(), lr=5e-5, weight_decay=1e-5)
        seed = torch.zeros(1, dtype=torch.int64, device=dev)
        if name == "vasnet":
            m.graph_seed = seed
        def step(captured=False):
            # (captured: the trainers' graph form -- the Adam kernel leaves the gradient bucket zero, so no fill kernel opens the step)
            if not captured:
                opt.zero_grad(zeroed_by_step=True)
            loss = SegmentMseMeanFunction.apply(m.score_packed(x2, [T]), target, sb, 1.0)      # one video (VASNetTrainer._single_video_step)
            loss.backward(gradient=_k_one(loss))
            opt.step(grad_scale=1.0, max_norm=5.0 if name == "dsn" else None, zero_grad=captured)
            seed.add_(1)
        tt_eager = timed(step, n=iters // 2, warm=50)
        opt.zero_grad(zeroed_by_step=True)
        tt_graph = timed(graphed(lambda: step(True)), n=iters // 2, warm=50)
        kernels.health_check()
        rec = lambda t: dict(us_per_video=round(t * 1e6, 1), frames_per_s=round(T / t, 1))
        out[name] = dict(score_eager=rec(t_eager), train_step_eager=rec(tt_eager), train_step_graph=rec(tt_graph))
    out["note"] = (f"one video per call (T={T}, D={D}, fp32), features resident in HBM; score = model.forward((T,1,D)); train step = zero_grad + "
                   "forward + per-video MSE + backward + fused Adam (DSN: + grad-norm clip), dropout on; train_step_graph = the step captured once "
                   "into a HIP graph and replayed (what VASNetTrainer does from its second epoch on)")
    return out


def trainer_test_leg(dev, D=1024, n_videos=50):
    """SURVEY 8d, the metric as the reference defines it: the `Trainer.test`-style inference loop END TO END -- packed scoring of the
    fold's 50 test videos (features cached in HBM after the first call), upsample + segment means + Spearman on the device, one small
    D2H, knapsack key-shot selection + F-scores on the host threads -- wall clock per call and frames/s."""
    from summarizer_amd.models.vasnet import VASNetTrainer
    from summarizer_amd.utils.datasets import synthetic_dataset
    from summarizer_amd.utils.hps import make_hps
    ds = synthetic_dataset(n_videos, seed=11, D=D, t_range=(150, 320), n_users=20)
    keys = list(ds.keys())
    hps = make_hps(ds, [{"train_keys": [], "test_keys": keys}], epochs=1, extra_params={})
    torch.manual_seed(1234)
    tr = VASNetTrainer(hps, hps.splits_files[0]).reset()
    frames = int(sum(ds[k]["features"].shape[0] for k in keys))
    for _ in range(10):
        res = tr.test(0)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        res = tr.test(0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    assert all(np.isfinite(v) for v in (res[0], res[1][0], res[1][1]))
    return dict(ms_per_call=round(dt * 1e3, 3), frames_per_s=round(frames / dt, 1), videos=n_videos, frames=frames,
                note="Trainer.test(fold) on 50 synthetic TVSum-shaped videos: packed VASNet scoring + device evaluation tail (upsample, segment "
                     "means, Spearman) + host knapsack / F-scores; features resident in HBM after the first call")


def stream_leg(model, x, lens, dev, steps=60):
    """SURVEY 8d's PCIe-inclusive variant (never `value`): features start in pageable host memory and scores end in host memory every
    step -- native threaded pack into pinned staging, one H2D, packed scoring, one D2H, three slots in flight (ingest.StreamingScorer)."""
    from summarizer_amd.ingest import StreamingScorer
    xh = x.cpu().numpy()
    off = np.concatenate([[0], np.cumsum(lens)])
    vids = [(i, xh[off[i]:off[i + 1]]) for i in range(len(lens))]
    frames = int(sum(lens))
    scorer = StreamingScorer(model, max_frames=frames, depth=3)
    def run(n):
        def feed():
            for _ in range(n):
                yield from vids
        for _ in scorer.score(feed()):
            pass
    run(5)
    t0 = time.perf_counter()
    run(steps)
    dt = (time.perf_counter() - t0) / steps
    return dict(ms_per_step=round(dt * 1e3, 3), frames_per_s=round(frames / dt, 1), h2d_bytes_per_step=int(frames * x.shape[1] * 4),
                note="host -> host: fp32 features from pageable host memory, scores back to host memory, every step")


def recurrent_legs(x, lens, dev, frames, with_cpu=True):
    """BASELINE config 3 on the headline batch: DSN (BiLSTM 1024 -> 2 x 256, dsn.py:38-47) scoring, MSE training step, and sLSTM
    (2-layer BiLSTM, H = 1024, sumgan.py:23-46) scoring, each scoring leg in exact fp32 and the split-bf16 modes.  A recurrence is a
    chain of T dependent steps: next to frames/s a leg reports the time of ONE step of its recurrence launch (HIP events on the launch
    stream, a separate pass) against the step's MFMA floor -- the padded product of one step on the CUs of one team at the nominal clock --
    and (scoring, fp32) the oracle port (torch's own nn.LSTM) on the host cores: a bounded sample, best of a short thread sweep."""
    from summarizer_amd import _lib, kernels as _k
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.models.sumgan import sLSTM
    from summarizer_amd.training import FlatAdam
    from summarizer_amd.autograd import SegmentMseMeanFunction
    lib = _lib.load()
    t_max = max(lens)
    out = {}

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(r).all())
        return (time.perf_counter() - t0) / n

    def rec_us(fn, n):          # the recurrence launches of a call (tag SUMK_PROF_LSTM_REC), microseconds per launch
        lib.sumk_prof_read(_lib.PROF_LSTM_REC, None, None, 1)
        lib.sumk_prof_enable(1 << _lib.PROF_LSTM_REC)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        lib.sumk_prof_enable(0)
        ms = C.c_double(0); cnt = C.c_int64(0)
        lib.sumk_prof_read(_lib.PROF_LSTM_REC, C.byref(ms), C.byref(cnt), 1)
        return ms.value / max(cnt.value, 1) * 1e3

    cu_f32 = FP32_MFMA_PEAK_TFLOPS * 1e12 / 256      # FLOP/s of one CU at the nominal clock
    cu_b16 = BF16_MFMA_PEAK_TFLOPS * 1e12 / 256

    def score_leg(m, prec, n, layers, floor_us, what):
        m.precision = prec
        with torch.no_grad():
            dt = timed(lambda: m.score_packed(x, lens), n)
            us = rec_us(lambda: m.score_packed(x, lens), max(2, n // 2))
        m.precision = "fp32"
        return dict(ms_per_step=round(dt * 1e3, 4), frames_per_s=round(frames / dt, 1), recurrence_launch_us=round(us, 1),
                    recurrence_us_per_step=round(us / t_max, 2), mfma_floor_us_per_step=round(floor_us, 2), frac_of_floor=round(floor_us / (us / t_max), 3),
                    us_per_recurrence_step=round(dt * 1e6 / (layers * t_max), 2), note=what)

    torch.manual_seed(1234)
    dsn = DSN(input_size=x.shape[1]).to(dev).eval()
    # one step of one team (32 CUs): 16 video rows (13 used) x 1024 gate columns x 256, exact fp32 MFMA
    dsn_floor = 2.0 * 16 * 1024 * 256 / (32 * cu_f32) * 1e6
    out["dsn_score_mode"] = score_leg(dsn, "fp32", 20, 1, dsn_floor,
        f"DSN scoring, 50 videos packed: input projection GEMM + one persistent bidirectional recurrence of {t_max} dependent steps + head; the step is a hand-off "
        "latency chain (flag-in-data publish -> poll -> 16 fp32 MFMAs per wave -> cell update), not a throughput: the floor is quoted for scale")
    out["dsn_score_bf16x6_mode"] = score_leg(dsn, "bf16x6", 20, 1, dsn_floor,
        "input projection in the fp32-grade bf16x6 arithmetic, computed INSIDE the persistent recurrence (csrc/lstm.hip, lstm_persist_proj_kernel: the step's x rows x "
        "the member's W_ih planes start the accumulators of the recurrent product while the wave would wait for the hand-off; no G, no GEMM launch); recurrent product exact fp32")
    out["dsn_score_bf16x3_mode"] = score_leg(dsn, "bf16x3", 20, 1, dsn_floor, "the same with two planes (bf16x3)")
    if with_cpu:
        try:
            out["dsn_score_mode"]["cpu_baseline"] = cpu_baseline(lens, x.shape[1], budget_s=4.0, kind="dsn", threads=(8, 16, 32))
            out["dsn_score_mode"]["cpu_baseline"].pop("parity_max_abs_diff_vs_port", None)
        except Exception as e:          # noqa: BLE001
            out["dsn_score_mode"]["cpu_baseline"] = dict(error=f"{type(e).__name__}: {e}"[:200])
    dsn.train()
    opt = FlatAdam(dsn.parameters(), lr=1e-5, weight_decay=1e-5)
    target = torch.rand(frames, device=dev)
    sb = _k.SeqBatch.get(lens, dev)

    def train_step():
        opt.zero_grad(zeroed_by_step=True)
        loss = SegmentMseMeanFunction.apply(dsn.score_packed(x, lens), target, sb, 1.0 / len(lens))
        loss.backward(gradient=_k_one(loss))
        opt.step(grad_scale=1.0, zero_grad=True)
        return loss.detach()
    dt = timed(train_step, 10)
    out["dsn_train_mode"] = dict(ms_per_step=round(dt * 1e3, 4), frames_per_s=round(frames / dt, 1), us_per_recurrence_step=round(dt * 1e6 / (2 * t_max), 2),
                                 note="DSN MSE training step (forward + per-video MSE + BPTT + fused Adam); recurrence steps = forward + backward")
    del dsn, opt
    sl = sLSTM(input_size=x.shape[1]).to(dev).eval()
    # one step of one direction's team (128 CUs): 64 video rows (two 32-row tiles; 50 used) x 4096 gate columns x 1024
    sl_fl = 2.0 * 64 * 4096 * 1024
    out["slstm_score_mode"] = score_leg(sl, "fp32", 5, 2, sl_fl / (128 * cu_f32) * 1e6,
        f"sLSTM scoring (2 layers x {t_max} dependent steps, H = 1024): lstm_wide2_kernel -- exchange buffer laid out for the consumers, rows sorted by length "
        "(the second 32-row tile stops once fewer than 33 videos run: the floor quoted is the TWO-tile step), sharded step counter -- + 4 input projections")
    out["slstm_score_bf16x6_mode"] = score_leg(sl, "bf16x6", 5, 2, 6.0 * sl_fl / (128 * cu_b16) * 1e6,
        "input projections on operand planes AND the recurrent product in the fp32-grade bf16x6 arithmetic (three planes of W_hh, h split in registers, 6 bf16 MFMAs per product)")
    out["slstm_score_bf16x3_mode"] = score_leg(sl, "bf16x3", 5, 2, 3.0 * sl_fl / (128 * cu_b16) * 1e6, "the same with two planes (bf16x3)")
    if with_cpu:
        try:
            out["slstm_score_mode"]["cpu_baseline"] = cpu_baseline(lens, x.shape[1], budget_s=6.0, kind="slstm", threads=(16, 32))
            out["slstm_score_mode"]["cpu_baseline"].pop("parity_max_abs_diff_vs_port", None)
        except Exception as e:          # noqa: BLE001
            out["slstm_score_mode"]["cpu_baseline"] = dict(error=f"{type(e).__name__}: {e}"[:200])
    _k.health_check()
    return out


def transformer_legs(x, lens, dev, frames):
    """SURVEY 8 row f2 on the headline batch: the Transformer-encoder scorer (6 post-norm layers, 8 heads, transformer.py:74-103), exact
    fp32 and the fp32-grade bf16x6 arithmetic (every projection of the stack on the plane GEMM, csrc/gemm_pw.hip)."""
    from summarizer_amd.models.transformer import Transformer
    torch.manual_seed(1234)
    m = Transformer(input_size=x.shape[1]).to(dev).eval()
    flops = frames * 6 * (2 * 6 * x.shape[1] ** 2) + 2 * frames * x.shape[1] ** 2 + 6 * 4 * sum(t * t for t in lens) * x.shape[1]
    out = {}
    ref = None
    for prec, key in (("fp32", "transformer_score_mode"), ("bf16x6", "transformer_score_bf16x6_mode"), ("bf16x3", "transformer_score_bf16x3_mode")):
        m.precision = prec
        with torch.no_grad():
            for _ in range(3):
                s = m.score_packed(x, lens)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                s = m.score_packed(x, lens)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        assert bool(torch.isfinite(s).all())
        rec = dict(ms_per_step=round(dt * 1e3, 4), frames_per_s=round(frames / dt, 1), whole_path_tflops=round(flops / dt / 1e12, 1))
        if ref is None:
            ref = s
            rec["note"] = "Transformer-encoder scorer, 50 videos packed, exact fp32 MFMA (fraction of the 157.3 TFLOP/s fp32 peak: %.2f)" % (flops / dt / 157.3e12)
        else:
            rec["max_abs_diff_vs_fp32_scores"] = float((s - ref).abs().max())
            rec["note"] = (f"the same in {prec}: weights as cached bf16 planes, activations split once per projection or written as planes by the producing epilogue, "
                           "per-head attention on planes (heads of 128 columns, T <= 320)")
        out[key] = rec
    return out


def stress_leg(dev, steps=3):
    """BASELINE config 5 at one GPU's share: 8 sequences of T = 10 000 frames, D = 2048 (the attention matrix of one sequence is 400 MB: E
    is materialised, 3.2 GB), exact fp32 and the fp32-grade bf16x6 arithmetic.  Reports the whole-path fraction of the arithmetic's MFMA peak
    (the path is matrix-bound, not HBM-bound: ~10 D^2 + 4 T D FLOP per frame against ~30 KB of HBM traffic per frame)."""
    from summarizer_amd.models.vasnet import VASNet
    D, lens = 2048, [10000] * 8
    frames = sum(lens)
    torch.manual_seed(1234)
    model = VASNet(input_size=D).to(dev).eval()
    g = torch.Generator(device=dev); g.manual_seed(0)
    x = torch.randn(frames, D, device=dev, generator=g) * 0.05
    flops = frames * (10.0 * D * D + 2.0 * D) + 4.0 * sum(t * t for t in lens) * D
    out = {}
    ref = None
    for prec, peak, key in (("fp32", FP32_MFMA_PEAK_TFLOPS, "stress_mode"), ("bf16x6", BF16_MFMA_PEAK_TFLOPS / 6.0, "stress_bf16x6_mode")):
        model.precision = prec
        with torch.no_grad():
            s = model.score_packed(x, lens)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                s = model.score_packed(x, lens)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert bool(torch.isfinite(s).all())
        tf = flops / dt / 1e12
        out[key] = dict(ms_per_step=round(dt * 1e3, 3), frames_per_s=round(frames / dt, 1), steps=steps, whole_path_tflops=round(tf, 2),
                        roofline=dict(bound="mfma", achieved=round(tf, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(tf / peak, 4), traffic=None,
                                      note="whole step (all kernels) against the arithmetic's MFMA peak (bf16x6: 2.5 PFLOP/s / 6); 45 % of the FLOP are the (T x T) attention products"),
                        workload=f"vasnet score, S-stress (BASELINE config 5): 8 sequences/GPU, T=10000, D=2048, packed batch, {prec}")
        if ref is None:
            ref = s
        else:
            out[key]["max_abs_score_diff_vs_fp32"] = float((s - ref).abs().max())
    model.precision = "fp32"
    del x, model, s, ref
    torch.cuda.empty_cache()
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: THIS process has not touched the GPU yet (torch is imported, nothing is
    initialised), so it starts N fresh rank processes through torch.distributed.run as CHILDREN (never an exec), lets rank 0's JSON
    line through on the inherited stdout and exits with the launcher's code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--videos", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--notes", action="store_true",
                    help="keep every leg's free-text note / provenance fields in the JSON line (about 2x its length).  Default: numbers only, so that "
                         "the whole line fits the 8 KB of stdout tail the round driver keeps; profiles/ holds a --notes copy of the line")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the side legs (split-bf16 / folded modes, per-kernel pass, training leg): what profiling runs use")
    ap.add_argument("--no-probe", action="store_true", help="skip the MFMA rate probe (mfma_sustained): profiling runs, whose kernel table it would lead")
    ap.add_argument("--model", choices=["vasnet", "dsn", "slstm", "transformer", "sumgan"], default="vasnet",
                    help="headline = vasnet; dsn = BiLSTM 1024->2x256; slstm = SumGAN's 2-layer BiLSTM 1024->2x1024; "
                         "sumgan (--mode train only) = one SumGANTrainer video step: selector+encoder, decoder and "
                         "discriminator updates at the reference's default sizes (195 M parameters)")
    ap.add_argument("--graph", action="store_true", help="capture the step into a HIP graph after warm-up and time the replays "
                    "(single process; roofline events are not recorded inside a captured step)")
    ap.add_argument("--mode", choices=["score", "train", "reinforce", "stream"], default="score",
                    help="headline = score (frames scored/sec, features resident in HBM); train = MSE step; reinforce = DSN "
                         "REINFORCE step (BASELINE config 4); stream = PCIe-inclusive scoring: features start in pageable host "
                         "memory and scores end there (summarizer_amd/ingest.py) -- never the headline value")
    ap.add_argument("--stage-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="--mode stream only: bf16 = the native packer converts while it copies, half the PCIe bytes, LOSSY (scores of bf16(features))")
    ap.add_argument("--workload", choices=["tvsum", "stress"], default="tvsum",
                    help="tvsum = S-TVSum headline; stress = BASELINE config 5: T=10000, D=2048, 8 sequences per GPU")
    ap.add_argument("--fold-vo", action="store_true",
                    help="VASNet scoring with Wo.Wv folded (VASNet(fold_vo=True), opt-in inference mode): a non-headline line of its own, for profiling")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16x6", "bf16"], default="fp32",
                    help="GEMM arithmetic: fp32 = exact fp32 MFMA (headline); bf16x6 / bf16x3 = fp32 operands split into 3 / 2 bf16 "
                         "planes, 6 / 3 bf16 MFMAs per product, fp32 accumulate (fp32-grade / ~1e-5 on scores); bf16 = plain bf16 "
                         "operands, 1 MFMA per product: the mixed-precision TRAINING mode of BASELINE config 2 (use with --mode train)")
    args = ap.parse_args()
    if args.fold_vo:
        args.headline_only = True          # (the side legs toggle fold_vo themselves)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; pass --gpus {world} "
              "(or run plain `python bench.py --gpus N`, which starts its own N ranks)", file=sys.stderr, flush=True)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # SUMK_BENCH_ONE_GPU=1 (tests only): every rank on cuda:0 over gloo, to exercise the multi-rank control flow on a 1-GPU box
    one_gpu = os.environ.get("SUMK_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ranks_seen = 1
    if dist is not None:
        # first contact with the collective backend (RCCL over xGMI on a multi-GPU node): a SUM all-reduce of ones must see every rank
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, f"all-reduce of ones saw {ranks_seen} ranks, expected {world}"

    import recipes as R
    from summarizer_amd import _lib
    from summarizer_amd.models.vasnet import VASNet
    lib = _lib.load()

    if args.model == "sumgan":
        return bench_sumgan(args, dev, rank, world, dist)
    D = 1024
    lens = tvsum_lens(args.videos)
    if args.workload == "stress":
        D, lens = 2048, [10000] * 8
    frames = int(sum(lens))
    torch.manual_seed(1234)
    if args.model == "vasnet":
        model = VASNet(input_size=D).to(dev)
    elif args.model == "transformer":
        from summarizer_amd.models.transformer import Transformer
        model = Transformer(input_size=D).to(dev)
    elif args.model == "slstm":
        from summarizer_amd.models.sumgan import sLSTM
        model = sLSTM(input_size=D).to(dev)
    else:
        from summarizer_amd.models.dsn import DSN
        model = DSN(input_size=D).to(dev)
    model.precision = args.precision
    if args.fold_vo:
        assert args.model == "vasnet" and args.mode == "score", "--fold-vo: VASNet scoring only"
        model.fold_vo = True
    model.train(args.mode not in ("score", "stream"))
    if args.workload == "stress":
        g = torch.Generator(device=dev); g.manual_seed(rank)
        x = torch.randn(frames, D, device=dev, generator=g) * 0.05
    else:
        x = torch.from_numpy(np.concatenate([R.features(T, 1, D, 1000 * rank + i)[:, 0, :] for i, T in enumerate(lens)])).to(dev)
    if args.mode == "train":
        # one optimiser step per packed batch: forward + MSE to a random target + backward + flat-bucket Adam
        # (+ one gradient all-reduce under torch.distributed)
        from summarizer_amd.training import FlatAdam
        opt = FlatAdam(model.parameters(), lr=5e-5, weight_decay=1e-5, comm_dtype=torch.bfloat16 if args.precision == "bf16" else None)
        target = torch.rand(frames, device=dev)
        from summarizer_amd import kernels as _k
        from summarizer_amd.autograd import SegmentMseMeanFunction
        sb_t = _k.SeqBatch.get(lens, dev)
        def run_step():
            opt.zero_grad(zeroed_by_step=True)
            loss = SegmentMseMeanFunction.apply(model.score_packed(x, lens), target, sb_t, 1.0 / len(lens))   # the trainers' loss: mean over videos of nn.MSELoss per video
            loss.backward(gradient=_k_one(loss))
            opt.step(grad_scale=opt.all_reduce_grads(), zero_grad=True)      # (the Adam kernel leaves the bucket zero: the next zero_grad() is free)
            return loss.detach()
    elif args.mode == "reinforce":
        assert args.model == "dsn", "--mode reinforce is the DSN trainer step"
        run_step, opt = make_reinforce_step(model, x, lens, dev)
    elif args.mode == "stream":
        # host -> host: every step ships the batch again (packed pinned staging, one H2D, packed scoring, one D2H), two slots deep
        from summarizer_amd.ingest import StreamingScorer
        xh = x.cpu().numpy()
        off = np.concatenate([[0], np.cumsum(lens)])
        vids = [(i, xh[off[i]:off[i + 1]]) for i in range(len(lens))]
        scorer = StreamingScorer(model, max_frames=frames, depth=3, stage_dtype=args.stage_dtype)
        last = [None]
        def run_steps(n):
            def feed():
                for _ in range(n):
                    yield from vids
            for _, sc in scorer.score(feed()):
                last[0] = sc
            return torch.from_numpy(last[0])
        run_step = None
    else:
        def run_step():
            with torch.no_grad():
                return model.score_packed(x, lens)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    graph = None
    if args.graph:
        # --graph: the whole step (forward, loss, autograd backward, optimiser; Bernoulli draws through torch's graph-safe Philox
        # state) is captured once into a HIP graph and REPLAYED: the launch-bound phases (the ~80 small kernels of a REINFORCE step)
        # no longer wait for the host.  Every libsumk entry point only enqueues on the caller's stream, so it captures as is.
        assert run_step is not None and dist is None, "--graph: single-process step modes only"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                run_step()                      # allocator pools, workspace cache, one-time function attributes
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_out = run_step()
        eager_step = run_step
        def run_step():
            graph.replay()
            return graph_out

    if run_step is None:
        s = run_steps(args.warmup)
    else:
        for _ in range(args.warmup):
            s = run_step()
    barrier()
    lib.sumk_prof_read(_lib.PROF_GEMM_QKV, None, None, 1)
    lib.sumk_prof_enable(1 << _lib.PROF_GEMM_QKV)      # only the dominant kernel is bracketed with events inside the timed region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if run_step is not None else []
    t0 = time.perf_counter()
    if run_step is None:
        s = run_steps(args.steps)
    else:
        marks[0].record()
        for i in range(args.steps):
            s = run_step()
            marks[i + 1].record()          # per-step device time (median and p10 / p90), no host sync inside the loop
    barrier()
    t1 = time.perf_counter()
    step_ms = None
    if marks:
        per = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
        step_ms = dict(median=round(float(np.median(per)), 4), p10=round(float(np.percentile(per, 10)), 4),
                       p90=round(float(np.percentile(per, 90)), 4))
    lib.sumk_prof_enable(0)
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(s).all())

    ms = C.c_double(0); n = C.c_int64(0)
    lib.sumk_prof_read(_lib.PROF_GEMM_QKV, C.byref(ms), C.byref(n), 1)
    qkv_flops = 2.0 * frames * (3 * D) * D
    roof = None
    if n.value > 0:
        avg_s = ms.value / n.value / 1e3
        ach = qkv_flops / avg_s / 1e12
        # bf16x3 issues 3 dense-bf16 MFMA flops per algorithmic flop: its ceiling is the bf16 peak / 3
        peak = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else round(BF16_MFMA_PEAK_TFLOPS / {"bf16": 1.0, "bf16x3": 3.0, "bf16x6": 6.0}[args.precision], 1)
        # `traffic` (HBM bytes per launch) cannot be read from inside this process: it comes from separate rocprofv3 --pmc passes of
        # THIS launch shape kept under profiles/ (scripts/pmc_pass.sh + pmc_to_json.py); everything derived from that file sits
        # under `pmc_reference`, away from the numbers measured live in this run.
        traffic, pmc_ref = None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and args.workload == "tvsum" and args.model == "vasnet" and args.videos == 50 and args.precision == "fp32":
            import hashlib
            raw = open(tp, "rb").read()
            pmc = json.loads(raw)
            traffic = pmc.get("gemm_qkv_hbm_bytes_per_launch")
            c = pmc.get("counters_mean_per_launch", {})
            pmc_ref = dict(source=f"profiles/pmc_traffic.json sha256 {hashlib.sha256(raw).hexdigest()[:12]}: separate rocprofv3 --pmc passes "
                                  "(FETCH_SIZE doubled per the gfx950 note), NOT measured in this run",
                           kernel=pmc.get("kernel"), hbm_bytes_per_launch=traffic, algorithmic_bytes_per_launch=pmc.get("algorithmic_bytes_per_launch"))
            if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
                pmc_ref["mfma_busy_frac_in_pmc_run"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)
        roof = dict(bound="mfma", kernel="gemm_f32_kernel<128,NT> (QKV projection)" + ("" if args.precision == "fp32" else f" [{args.precision}]"),
                    achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                    traffic=traffic, traffic_source=("static: profiles/pmc_traffic.json (see pmc_reference)" if traffic else None),
                    avg_launch_us=round(avg_s * 1e6, 2), launches=int(n.value), flops_per_launch=qkv_flops)
        if pmc_ref:
            roof["pmc_reference"] = pmc_ref

    # the other GEMMs of the step, each against the same peak (a separate short pass: their event pairs stay out of the timed region)
    kern = None
    if args.model == "vasnet" and args.mode == "score" and run_step is not None and not args.headline_only:
        tags = {"qkt": _lib.PROF_GEMM_QKT, "alpha_v": _lib.PROF_GEMM_PV, "out_proj": _lib.PROF_GEMM_OPROJ, "k1": _lib.PROF_GEMM_K1}
        for t in tags.values():                      # one tag per pass: every extra event pair in a step costs the others ~10 us
            lib.sumk_prof_read(t, None, None, 1)
            lib.sumk_prof_enable(1 << t)
            for _ in range(10):
                run_step()
            torch.cuda.synchronize()
        lib.sumk_prof_enable(0)
        sq = float(sum(t * t for t in lens))
        fl = {"qkt": 2.0 * sq * D, "alpha_v": 2.0 * sq * D, "out_proj": 2.0 * frames * D * D, "k1": 2.0 * frames * D * D}
        pk = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else BF16_MFMA_PEAK_TFLOPS / {"bf16": 1.0, "bf16x3": 3.0, "bf16x6": 6.0}[args.precision]
        kern = {}
        for name, t in tags.items():
            ms2 = C.c_double(0); n2 = C.c_int64(0)
            lib.sumk_prof_read(t, C.byref(ms2), C.byref(n2), 1)
            if n2.value:
                us = ms2.value / n2.value * 1e3
                kern[name] = dict(avg_launch_us=round(us, 2), tflops=round(fl[name] / us / 1e6, 2), frac_of_peak=round(fl[name] / us / 1e6 / pk, 4))

    alt = alt6 = folded = folded6 = folded3 = None
    if args.model == "vasnet" and args.mode == "score" and args.precision == "fp32" and not args.headline_only:
        # every rank runs the side legs, so ranks stay in step; a failing side leg is reported in its field and must not cost the
        # headline line (nor leave the other ranks waiting at the barrier below)
        def _leg(fn, *a):
            try:
                return fn(*a)
            except Exception as e:          # noqa: BLE001
                return dict(error=f"{type(e).__name__}: {e}"[:300])
        alt = _leg(alt_precision_leg, model, x, lens, s, args.steps, frames)
        alt6 = _leg(alt_precision_leg, model, x, lens, s, args.steps, frames, "bf16x6")
        folded = _leg(folded_leg, model, x, lens, s, args.steps, frames)
        folded6 = _leg(folded_leg, model, x, lens, s, args.steps, frames, "bf16x6")   # the fastest mode inside the fp32 tolerances
        folded3 = _leg(folded_leg, model, x, lens, s, args.steps, frames, "bf16x3")   # the fastest mode inside the 1e-4 parity gate
        barrier()
    # data-parallel TRAINING leg on the same batch (every rank): forward + MSE + backward + the flat-bucket gradient all-reduce +
    # fused Adam.  Scoring has no data-path collective, so this is what makes a multi-GPU run of this script exercise RCCL.
    train_leg = train_leg_bf16 = reinforce_leg = one_video_leg = None
    if args.model == "vasnet" and args.mode == "score" and args.workload == "tvsum" and not args.headline_only:
        from summarizer_amd.training import FlatAdam
        from summarizer_amd import kernels as _k
        from summarizer_amd.autograd import SegmentMseMeanFunction
        sb_t = _k.SeqBatch.get(lens, dev)
        target = torch.rand(frames, device=dev)

        def timed_steps(step_fn, n_train=10):
            """seconds per step, max over ranks, barriers on both sides"""
            for _ in range(3):
                l = step_fn()
            barrier()
            tt0 = time.perf_counter()
            for _ in range(n_train):
                l = step_fn()
            barrier()
            tel = time.perf_counter() - tt0
            if dist is not None:
                t = torch.tensor([tel], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tel = float(t.item())
            assert bool(torch.isfinite(l))
            return tel / n_train

        def allreduce_alone_us(numel, dtype, n=20):
            """the step's gradient exchange by itself: one SUM all-reduce of a bucket of that size, timed outside any step (us, max over ranks)"""
            if dist is None:
                return None
            buf = torch.zeros(numel, dtype=dtype, device=dev)
            for _ in range(5):
                dist.all_reduce(buf)
            barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                dist.all_reduce(buf)
            barrier()
            t = torch.tensor([(time.perf_counter() - t0) / n], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return round(float(t.item()) * 1e6, 1)

        def decompose(rec, ms_step, ms_local, alone_us):
            """what a SCALE run needs to read a leg: the exchange alone, the step without it, how much of it the step hid"""
            if dist is None:
                return rec
            exposed = max(0.0, (ms_step - ms_local) * 1e3)
            rec.update(allreduce_alone_us=alone_us, ms_per_step_without_allreduce=round(ms_local, 4), exposed_comm_us=round(exposed, 1),
                       overlap_hidden_us=round(max(0.0, alone_us - exposed), 1))
            return rec

        def run_train_leg(precision):
            """exact fp32, or the mixed-precision mode of BASELINE config 2 (bf16 products, bf16 gradient bucket over the all-reduce).
            Under data parallelism the exchange is issued as VASNetTrainer issues it: the tail of the bucket (Wo, k1, k2: final halfway
            through the backward pass) on a side stream under the attention backward, the head after it."""
            model.train(); model.precision = precision
            opt = FlatAdam(model.parameters(), lr=5e-5, weight_decay=1e-5, comm_dtype=torch.bfloat16 if precision == "bf16" else None)
            opt.broadcast()
            tail_from = opt.tail_offset(model.attention_head_projection.weight) if dist is not None else None
            model.tail_grads_ready_event = torch.cuda.Event() if dist is not None else None
            def train_step(reduce=True):
                opt.zero_grad(zeroed_by_step=True)
                loss = SegmentMseMeanFunction.apply(model.score_packed(x, lens), target, sb_t, 1.0 / len(lens))
                loss.backward(gradient=_k_one(loss))
                if reduce and tail_from is not None:
                    opt.reduce_tail_async(tail_from, model.tail_grads_ready_event)
                opt.step(grad_scale=opt.all_reduce_grads() if reduce else 1.0 / world, zero_grad=True)
                return loss.detach()
            n_train = 10
            sec = timed_steps(train_step, n_train)
            sec_local = timed_steps(lambda: train_step(False), n_train) if dist is not None else sec
            model.tail_grads_ready_event = None
            model.eval(); model.precision = "fp32"
            bytes_per_elem = 2 if precision == "bf16" else 4
            rec = dict(frames_per_s=round(frames * world / sec, 1), ms_per_step=round(sec * 1e3, 4), steps=n_train,
                       allreduce_bytes_per_step=int(opt.flat_grad.numel() * bytes_per_elem) if world > 1 else 0,
                       collectives_per_step=2 if world > 1 else 0,
                       note=(f"{precision}; forward + per-video MSE + backward + the all-reduce of the flat gradient bucket in two pieces (tail early on a "
                             "side stream, head after the backward; RCCL when world > 1) + fused Adam"))
            return decompose(rec, sec * 1e3, sec_local * 1e3,
                             allreduce_alone_us(opt.flat_grad.numel(), torch.bfloat16 if precision == "bf16" else torch.float32))
        def collective_leg(fn, *a):
            """A side leg that contains collectives must not cost the headline line: every rank catches its own exception, then the
            ranks agree (MIN all-reduce of an ok flag) whether the leg counts -- an error that hits all ranks alike (the usual kind) is
            reported in the leg's field and the line is still printed."""
            err, res = None, None
            try:
                res = fn(*a)
            except Exception as e:          # noqa: BLE001
                err = f"{type(e).__name__}: {e}"[:300]
            if dist is not None:
                flag = torch.tensor([0.0 if err else 1.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag.item()) == 0.0 and err is None:
                    err = "failed on another rank"
            return dict(error=err) if err else res
        train_leg = collective_leg(run_train_leg, "fp32")
        train_leg_bf16 = collective_leg(run_train_leg, "bf16")
        train_leg_x6 = collective_leg(run_train_leg, "bf16x6")      # the same step in the fp32-grade split arithmetic: one figure, inside the fp32 leg
        if isinstance(train_leg, dict) and isinstance(train_leg_x6, dict) and "ms_per_step" in train_leg and "ms_per_step" in train_leg_x6:
            train_leg["bf16x6_ms_per_step"] = train_leg_x6["ms_per_step"]

        def run_one_video_leg():
            """The trainers' DEFAULT schedule under data parallelism: batch_videos = 1 -- every rank steps on ONE video (vasnet.py:193-212
            scaled out by video), the 21 MB bucket all-reduced in two pieces every step.  The step is ~0.3 ms of compute, so this is the leg
            on which the exchange shows (predicted 0.55-0.85 at 8 GPUs); `training.choose_batch_videos` is what a
            caller who wants >= 0.9 uses (extra_params batch_videos=auto)."""
            from summarizer_amd.training import predicted_dp_efficiency
            model.train(); model.precision = "fp32"
            opt = FlatAdam(model.parameters(), lr=5e-5, weight_decay=1e-5)
            opt.broadcast()
            tail_from = opt.tail_offset(model.attention_head_projection.weight) if dist is not None else None
            model.tail_grads_ready_event = torch.cuda.Event() if dist is not None else None
            T1 = lens[0]
            x1, t1, sb1 = x[:T1].contiguous(), target[:T1].contiguous(), _k.SeqBatch.get([T1], dev)
            def step1(reduce=True):
                opt.zero_grad(zeroed_by_step=True)
                loss = SegmentMseMeanFunction.apply(model.score_packed(x1, [T1]), t1, sb1, 1.0 / world)
                loss.backward(gradient=_k_one(loss))
                if reduce and tail_from is not None:
                    opt.reduce_tail_async(tail_from, model.tail_grads_ready_event)
                opt.step(grad_scale=opt.all_reduce_grads(average=False) if reduce else 1.0, zero_grad=True)
                return loss.detach()
            n_train = 40
            sec = timed_steps(step1, n_train)
            sec_local = timed_steps(lambda: step1(False), n_train) if dist is not None else sec
            model.tail_grads_ready_event = None
            model.eval()
            rec = dict(frames_per_s=round(T1 * world / sec, 1), ms_per_step=round(sec * 1e3, 4), steps=n_train, videos_per_rank_per_step=1, frames_per_video=T1,
                       allreduce_bytes_per_step=int(opt.flat_grad.numel() * 4) if world > 1 else 0, collectives_per_step=2 if world > 1 else 0,
                       predicted_efficiency_at_this_world=predicted_dp_efficiency("vasnet", "fp32", max(world, 2), 1),
                       note="eager steps (the per-video HIP graphs of VASNetTrainer are single-process only); fp32; the all-reduce in two pieces as the trainer issues it")
            return decompose(rec, sec * 1e3, sec_local * 1e3, allreduce_alone_us(opt.flat_grad.numel(), torch.float32))
        one_video_leg = collective_leg(run_one_video_leg)

        def run_reinforce_leg():
            """BASELINE config 4: the DSN REINFORCE step, data-parallel by video (every rank its own 50 videos, one all-reduce of the
            10.5 MB flat gradient bucket per step, clip after the reduction, per-video baselines rank-local)."""
            from summarizer_amd.models.dsn import DSN
            torch.manual_seed(1234)
            dsn = DSN(input_size=D).to(dev).train()
            step, opt = make_reinforce_step(dsn, x, lens, dev)
            opt.broadcast()
            n_train = 10
            sec = timed_steps(step, n_train)
            sec_local = sec
            if dist is not None:          # the same step with the exchange left out (every rank then steps on its own gradient)
                sec_local = timed_steps(lambda: step(False), n_train)
            dsn.precision = "bf16x6"                                      # the same step with the projections' products in the fp32-grade split arithmetic
            sec_x6 = timed_steps(step, n_train)
            dsn.precision = "fp32"
            from summarizer_amd import kernels as _kk
            _kk.health_check()                                            # persistent recurrences: no hand-off timed out
            rec = dict(frames_per_s=round(frames * world / sec, 1), ms_per_step=round(sec * 1e3, 4), bf16x6_ms_per_step=round(sec_x6 * 1e3, 4), steps=n_train,
                       allreduce_bytes_per_step=int(opt.flat_grad.numel() * 4) if world > 1 else 0,
                       collectives_per_step=2 if world > 1 else 0,
                       note="DSN (BiLSTM 1024 -> 2 x 256) REINFORCE step: scores, 5 Bernoulli episodes, reward kernel, policy loss, "
                            "backward, the all-reduce of the flat gradient bucket in two pieces (RCCL when world > 1: [reverse direction | head] early on a side "
                            "stream under the forward direction's weight-gradient GEMMs, the rest after the backward), clip + fused Adam")
            return decompose(rec, sec * 1e3, sec_local * 1e3, allreduce_alone_us(opt.flat_grad.numel(), torch.float32))
        reinforce_leg = collective_leg(run_reinforce_leg)
    single = None
    if args.model == "vasnet" and args.mode == "score" and args.workload == "tvsum" and not args.headline_only and rank == 0:
        try:          # rank 0 only, no collectives inside: the other ranks wait at destroy_process_group
            single = single_video_leg(dev)
        except Exception as e:          # noqa: BLE001
            single = dict(error=f"{type(e).__name__}: {e}"[:300])
    e2e = stream = recurrent = stress = tf_legs = None
    if args.model == "vasnet" and args.mode == "score" and args.workload == "tvsum" and not args.headline_only and rank == 0:
        def _side(fn, *a):
            try:
                return fn(*a)
            except Exception as e:          # noqa: BLE001
                return dict(error=f"{type(e).__name__}: {e}"[:300])
        model.eval(); model.precision = "fp32"
        e2e = _side(trainer_test_leg, dev)
        stream = _side(stream_leg, model, x, lens, dev)
        recurrent = _side(recurrent_legs, x, lens, dev, frames, not args.no_cpu_baseline)
        stress = _side(stress_leg, dev)
        tf_legs = _side(transformer_legs, x, lens, dev, frames)
    if dist is not None and args.model == "vasnet" and args.mode == "score" and args.workload == "tvsum" and not args.headline_only:
        barrier()          # the rank-0-only legs above take a few seconds: the other ranks wait here, not inside destroy_process_group
    if rank == 0:
        flops_frame = 10 * D * D + 4 * (sum(t * t for t in lens) / frames) * D + 2 * D
        out = dict(metric="frames scored/sec (T x 1024)", value=round(frames * world * args.steps / elapsed, 1),
                   unit="frames/s", n_gpus=world, rccl_ranks_seen=ranks_seen if (dist is not None and not one_gpu) else None,
                   ranks_seen=ranks_seen, collective_backend=(None if dist is None else dist.get_backend()),
                   steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 4), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f32" if args.precision == "fp32" else ("bf16 products, f32 accumulate/storage/master weights" if args.precision == "bf16" else f"f32 storage/accumulate, {args.precision} split products"),
                   data="synthetic",
                   config=dict(workload=(f"{args.model} {args.mode}, S-TVSum: {args.videos} videos/GPU, T~U(150,320) (sum {frames}), D=1024, packed batch"
                                         if args.workload == "tvsum" else
                                         f"{args.model} {args.mode}, S-stress (BASELINE config 5): 8 sequences/GPU, T=10000, D=2048, packed batch"),
                               frames_per_step_per_gpu=frames, parallelism=f"video-sharded x{world}"),
                   whole_path_tflops=round(frames * world * args.steps / elapsed * flops_frame / 1e12, 2),
                   roofline=roof)
        if graph is not None:
            out["config"]["hip_graph"] = "step captured once, timed region = graph replays"
        sus = None if args.no_probe else mfma_sustained()
        if sus:
            out["mfma_sustained"] = sus
            if roof and roof.get("bound") == "mfma":
                key = "f32_32x32x2" if args.precision == "fp32" else "bf16_32x32x16"
                div = {"fp32": 1.0, "bf16": 1.0, "bf16x3": 3.0, "bf16x6": 6.0}[args.precision]
                roof["frac_of_sustained"] = round(roof["achieved"] / (sus[key] / div), 4)
        if step_ms is not None:
            out["step_ms_device_events"] = step_ms
        if kern:
            out["gemm_kernels"] = kern
        if train_leg:
            out["train_step_mode"] = train_leg
            out["train_step_bf16_mode"] = train_leg_bf16
            out["dsn_reinforce_step_mode"] = reinforce_leg
            out["dp_one_video_per_rank_mode"] = one_video_leg
        if args.fold_vo:
            out["config"]["workload"] += ", fold_vo=True (NOT the headline configuration: 18 % fewer executed FLOPs)"
        if args.model != "vasnet" or args.mode != "score" or args.workload != "tvsum":
            out["note"] = "non-headline mode: roofline/whole_path figures refer to the VASNet scoring FLOP model"
        if args.mode == "stream":
            out["note"] = ("PCIe-inclusive: features start in pageable host memory and scores end in host memory every step "
                           "(native threaded pack into pinned staging + one H2D + packed scoring + one D2H, 3 slots in flight)")
        if alt is not None:
            out["bf16x6_mode"] = alt6
            out["bf16x3_mode"] = alt
            out["folded_vo_mode"] = folded
            out["folded_vo_bf16x6_mode"] = folded6
            out["folded_vo_bf16x3_mode"] = folded3
        if single is not None:
            out["single_video_mode"] = single
        if e2e is not None:
            out["trainer_test_mode"] = e2e
        if stream is not None:
            out["stream_mode"] = stream
        if recurrent is not None:
            if "error" in recurrent:
                out["dsn_score_mode"] = recurrent
            else:
                out.update(recurrent)
        if stress is not None:
            if "error" in stress:
                out["stress_mode"] = stress
            else:
                out.update(stress)
        if tf_legs is not None:
            if "error" in tf_legs:
                out["transformer_score_mode"] = tf_legs
            else:
                out.update(tf_legs)
        if world == 1 and not args.no_cpu_baseline and args.model in ("vasnet", "dsn", "slstm") and args.mode == "score" and args.workload == "tvsum":
            try:
                out["cpu_baseline"] = cpu_baseline(lens, D, kind=args.model, gpu_scores=s if args.precision != "bf16" else None, seed_base=1000 * rank)
                # the headline batch against the oracle port, every video (gate 1e-4): what `value` times is what was checked
                out["parity_max_abs_diff_vs_port"] = out["cpu_baseline"].pop("parity_max_abs_diff_vs_port")
                out["parity_gate"] = 1e-4
                if single is not None and "vasnet" in single and args.model == "vasnet":      # the same one-video-per-call pattern on the host cores
                    cpu = out["cpu_baseline"]["value"]
                    single["vasnet"]["score_vs_cpu_port"] = {k: round(single["vasnet"][k]["frames_per_s"] / cpu, 1) for k in ("score_eager",)}
            except Exception as e:          # noqa: BLE001
                out["cpu_baseline"] = dict(error=f"{type(e).__name__}: {e}"[:300])
        if not args.notes:
            out = _compact(out)
            out["notes"] = "per-leg notes and provenance: python bench.py --notes (a copy of that line: profiles/r06_bench_default_line_notes.json)"
        print(json.dumps(out), flush=True)
        pd = out.get("parity_max_abs_diff_vs_port")
        if pd is not None and not pd < out["parity_gate"]:
            print(f"bench.py: PARITY FAILURE: max |score - port| = {pd} on the timed batch (gate {out['parity_gate']})", file=sys.stderr, flush=True)
            sys.exit(3)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
