"""Device-side evaluation tail (csrc/evaldev.hip): upsample + float32 segment means + Spearman on the GPU with the
scores still in HBM, key-shot selection / F-scores finished on the host -- against the all-host native tail (which test_host_eval.py
holds bit-exact to the numpy specification and to the reference's goldens)."""
import numpy as np
import pytest
import torch

import recipes as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", ["knapsack", "rank"])
def test_device_tail_equals_host_tail(method):
    from summarizer_amd.utils import eval as E
    from summarizer_amd.utils import eval_native
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    vids, scores, lens = [], [], []
    for i, (T, U) in enumerate([(300, 15), (1, 3), (97, 20), (650, 18), (33, 5), (320, 32), (150, 1)]):
        v = R.synthetic_video(T, 8100 + i, n_users=U)
        s = rng.random(T).astype(np.float32)
        if i == 2:
            s[10:40] = s[10]                                   # long runs of EQUAL step scores: tie groups across pick intervals
        if i == 4:
            s[:] = 0.25                                        # every frame tied
        vids.append(eval_native.prepare_video(v["n_frames"], v["picks"], v["change_points"], v["n_frame_per_seg"], v["user_summary"],
                                              E.rank_users(v["user_scores"])))
        scores.append(s); lens.append(T)
        assert eval_native.device_ready(vids[-1])
    want = eval_native.evaluate_batch(vids, scores, 0.15, method, want_summaries=True, n_threads=3)
    packed = torch.from_numpy(np.concatenate(scores)).to(dev)
    got = eval_native.evaluate_batch_device(vids, packed, lens, 0.15, method, want_summaries=True, n_threads=3)
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-12)          # Spearman: float64, different summation order
    np.testing.assert_array_equal(got[1], want[1])                           # F-scores: bit-identical (same integers into the knapsack)
    np.testing.assert_array_equal(got[2], want[2])
    for a, b in zip(got[3], want[3]):
        np.testing.assert_array_equal(a, b)
    # the batch's descriptors, device buffers and pinned buffers are kept between calls (Trainer.test asks for the same fold every time):
    # other scores for the same videos, then a sub-batch of them, still give the host tail's numbers
    scores2 = [rng.random(T).astype(np.float32) for T in lens]
    want2 = eval_native.evaluate_batch(vids, scores2, 0.15, method, n_threads=3)
    got2 = eval_native.evaluate_batch_device(vids, torch.from_numpy(np.concatenate(scores2)).to(dev), lens, 0.15, method, n_threads=3)
    np.testing.assert_allclose(got2[0], want2[0], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got2[1], want2[1]); np.testing.assert_array_equal(got2[2], want2[2])
    assert not np.array_equal(got2[0], got[0])
    sub = [0, 3, 5]
    want3 = eval_native.evaluate_batch([vids[i] for i in sub], [scores2[i] for i in sub], 0.15, method, n_threads=3)
    got3 = eval_native.evaluate_batch_device([vids[i] for i in sub], torch.from_numpy(np.concatenate([scores2[i] for i in sub])).to(dev),
                                             [lens[i] for i in sub], 0.15, method, n_threads=3)
    np.testing.assert_allclose(got3[0], want3[0], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got3[1], want3[1]); np.testing.assert_array_equal(got3[2], want3[2])


def test_device_tail_declines_what_it_does_not_cover():
    from summarizer_amd.utils import eval as E
    from summarizer_amd.utils import eval_native
    v = R.synthetic_video(60, 8200, n_users=4)
    ok = eval_native.prepare_video(v["n_frames"], v["picks"], v["change_points"], v["n_frame_per_seg"], v["user_summary"], E.rank_users(v["user_scores"]))
    assert eval_native.device_ready(ok)
    shuffled = eval_native.prepare_video(v["n_frames"], v["picks"][::-1].copy(), v["change_points"], v["n_frame_per_seg"], v["user_summary"],
                                         E.rank_users(v["user_scores"]))
    assert not eval_native.device_ready(shuffled)                            # picks not ascending -> host tail
    assert not eval_native.device_ready(eval_native.prepare_video(v["n_frames"], v["picks"]))          # no change points / ranks
