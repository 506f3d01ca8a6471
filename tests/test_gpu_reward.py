"""GPU parity: DSN reward kernel vs golden values from the real DSNTrainer.compute_reward and the numpy oracle."""
import numpy as np
import pytest
import torch

import recipes as R
from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_reward_goldens_and_batch():
    from oracle import reward_np
    from summarizer_amd import kernels
    dev = torch.device("cuda:0")
    g = load_golden("reward")
    for ci in range(5):
        seq, act = g[f"c{ci}/seq"], g[f"c{ci}/actions"]
        T = seq.shape[0]
        sb = kernels.SeqBatch.get([T], dev)
        x = torch.from_numpy(seq[:, 0, :].copy()).to(dev)
        a = torch.from_numpy(act.reshape(1, T).copy()).to(dev)
        for far in (0, 1):
            got = kernels.dsn_reward(x, sb, a, far_sim=bool(far)).item()
            ref = float(g[f"c{ci}/reward_far{far}"])
            if np.isnan(ref):           # one pick: the reference raises IndexError; oracle-defined value
                ref = float(reward_np.compute_reward(seq, act, far_sim=bool(far)))
            np.testing.assert_allclose(got, ref, atol=3e-5, rtol=1e-5, err_msg=f"case {ci} far={far}")
    # ragged batch, several episodes in one call, vs the oracle
    lens, D, E = [60, 1, 37, 130, 64], 96, 4
    xs = [R.features(T, 1, D, 300 + i) for i, T in enumerate(lens)]
    rng = np.random.default_rng(5)
    acts = (rng.random((E, sum(lens))) < 0.4).astype(np.float32)
    acts[1, :] = 0            # an episode with no picks anywhere -> reward 0
    sb = kernels.SeqBatch.get(lens, dev)
    got = kernels.dsn_reward(torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev), sb,
                             torch.from_numpy(acts).to(dev), temp_dist_thre=20).cpu().numpy()
    off = np.concatenate([[0], np.cumsum(lens)])
    for e in range(E):
        for s, x in enumerate(xs):
            ref = reward_np.compute_reward(x, acts[e, off[s]:off[s + 1]])
            np.testing.assert_allclose(got[e, s], ref, atol=3e-5, rtol=1e-5, err_msg=f"ep {e} video {s}")


def test_reward_at_baseline_size_vs_oracle():
    """BASELINE config 4 at size: sumk_dsn_reward on the S-TVSum batch (50 videos, T ~ U(150, 320), D = 1024) x 5 Bernoulli
    episodes -- reward_rows_kernel + the 64x64 Gram GEMMs on 12 003 frames -- against oracle/reward_np.compute_reward
    (dsn.py:185-236) video by video and episode by episode, default temporal threshold and far_sim."""
    from oracle import reward_np
    from summarizer_amd import kernels
    dev = torch.device("cuda:0")
    D, E = 1024, 5
    lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)]
    xs = [R.features(T, 1, D, 1000 + i) for i, T in enumerate(lens)]
    rng = np.random.default_rng(11)
    probs = rng.uniform(0.05, 0.6, sum(lens))
    acts = (rng.random((E, sum(lens))) < probs[None, :]).astype(np.float32)
    sb = kernels.SeqBatch.get(lens, dev)
    x = torch.from_numpy(np.concatenate([v[:, 0, :] for v in xs])).to(dev)
    off = np.concatenate([[0], np.cumsum(lens)])
    for far in (False, True):
        got = kernels.dsn_reward(x, sb, torch.from_numpy(acts).to(dev), far_sim=far, temp_dist_thre=20).cpu().numpy()
        assert got.shape == (E, len(lens))
        for s_, xv in enumerate(xs if not far else xs[:8]):          # (far_sim: the first 8 videos -- the oracle costs ~0.1 s per call)
            for e in range(E):
                ref = reward_np.compute_reward(xv, acts[e, off[s_]:off[s_ + 1]], far_sim=far)
                np.testing.assert_allclose(got[e, s_], ref, atol=3e-5, rtol=1e-5, err_msg=f"far={far} ep {e} video {s_}")


@pytest.mark.parametrize("E", [5, 16, 37])
def test_policy_loss_kernels_match_the_torch_ops_they_replace(E):
    """sumk_dsn_policy_loss_forward/backward (through PolicyLossFunction) against the element-wise torch formulation of
    dsn.py:113-140 -- Bernoulli.log_prob, per-video means, advantage product, length penalty, / E -- values and the gradient
    w.r.t. the probabilities, on a ragged batch that includes probabilities exactly at 0 and 1 (clamped: zero log-prob gradient)."""
    import torch
    from torch.distributions import Bernoulli
    from summarizer_amd import kernels
    from summarizer_amd.autograd import PolicyLossFunction
    dev = torch.device("cuda:0")
    lens, beta, eps = [70, 1, 33, 129, 5], 0.01, 0.5      # E = 37: more episodes than one register chunk (the reference takes any num_episodes, dsn.py:53)
    sb = kernels.SeqBatch.get(lens, dev)
    g = torch.Generator().manual_seed(3)
    p0 = torch.rand(sum(lens), generator=g) * 0.98 + 0.01
    p0[3] = 0.0; p0[40] = 1.0; p0[75] = 1e-9
    actions = (torch.rand(E, sum(lens), generator=g) < 0.4).float().to(dev)
    rewards = torch.rand(E, len(lens), generator=g).to(dev)
    base = torch.rand(len(lens), generator=g).to(dev)
    w = torch.rand(len(lens), generator=g).to(dev)                       # a non-trivial upstream gradient per video

    pa = p0.clone().to(dev).requires_grad_(True)
    lv = PolicyLossFunction.apply(pa, sb, actions, rewards, base, beta, eps)
    (lv * w).sum().backward()

    pb = p0.clone().to(dev).requires_grad_(True)
    dist = Bernoulli(pb, validate_args=False)
    ref = beta * (sb.segment_mean(pb) - eps) ** 2 - (sb.segment_mean(dist.log_prob(actions)) * (rewards - base)).sum(dim=0)
    ref = ref / float(E)
    (ref * w).sum().backward()
    torch.testing.assert_close(lv.detach(), ref.detach(), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(pa.grad, pb.grad, rtol=2e-4, atol=1e-6)
    # at p = 0 and p = 1 (clamped) only the length-penalty term is left: w_v * 2 beta (mean p - eps) / (E T_v)
    mp = sb.segment_mean(pb.detach())
    want = (w * 2 * beta * (mp - eps) / (E * torch.tensor(lens, dtype=torch.float32, device=dev)))[0]
    torch.testing.assert_close(pa.grad[[3, 40]], want.expand(2), rtol=1e-5, atol=0)


def test_segment_mse_kernels_match_torch():
    import torch
    from summarizer_amd import kernels
    from summarizer_amd.autograd import SegmentMseFunction
    dev = torch.device("cuda:0")
    lens = [70, 1, 300, 33, 129]
    sb = kernels.SeqBatch.get(lens, dev)
    g = torch.Generator().manual_seed(8)
    s0 = torch.rand(sum(lens), generator=g); y = torch.rand(sum(lens), generator=g).to(dev); w = torch.rand(len(lens), generator=g).to(dev)
    a = s0.clone().to(dev).requires_grad_(True); b = s0.clone().to(dev).requires_grad_(True)
    got = SegmentMseFunction.apply(a, y, sb); (got * w).sum().backward()
    ref = sb.segment_mean((b - y) ** 2); (ref * w).sum().backward()
    torch.testing.assert_close(got.detach(), ref.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-8)


def test_segment_mse_mean_equals_per_video_kernel_plus_torch_mean():
    """SegmentMseMeanFunction (the trainers' step loss in one launch each way: sumk_segment_mse_mean_*) against what it replaced --
    SegmentMseFunction + torch's mean / sum-and-divide and their autograd twins: the loss within one fp32 rounding of the sequential sum,
    the gradient BIT-IDENTICAL (2 (s - y) / T_v * (dloss * scale), the same operations in the same order), for a plain mean, a
    data-parallel divisor larger than the local video count, 2 000 videos (more than the block's 16 waves take in one round) and an
    upstream gradient other than one."""
    import torch
    from summarizer_amd import kernels
    from summarizer_amd.autograd import SegmentMseFunction, SegmentMseMeanFunction
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    for lens, n_total, up in (([70, 1, 300, 33, 129], 5, 1.0), ([240] * 50, 50, 1.0), ([17, 320, 5], 8, 1.0),
                              ([int(t) for t in torch.randint(1, 40, (2000,), generator=g)], 2000, 0.37)):
        sb = kernels.SeqBatch.get(lens, dev)
        s0 = torch.rand(sum(lens), generator=g); y = torch.rand(sum(lens), generator=g).to(dev)
        a = s0.clone().to(dev).requires_grad_(True); b = s0.clone().to(dev).requires_grad_(True)
        got = SegmentMseMeanFunction.apply(a, y, sb, 1.0 / n_total)
        assert got.shape == ()
        (got * up).backward()
        pv = SegmentMseFunction.apply(b, y, sb)
        ref = pv.mean() if n_total == len(lens) else pv.sum() / n_total
        (ref * up).backward()
        torch.testing.assert_close(got.detach(), ref.detach(), rtol=2e-6, atol=0)
        if n_total == len(lens) and up == 1.0:
            assert torch.equal(a.grad, b.grad)
        else:
            torch.testing.assert_close(a.grad, b.grad, rtol=3e-7, atol=0)
