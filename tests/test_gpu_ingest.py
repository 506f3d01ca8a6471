"""Streaming ingest (host features -> packed pinned staging -> one H2D -> packed scoring -> one D2H): results must equal
scoring resident features, bit for bit, whatever the batching."""
import numpy as np
import pytest
import torch

import recipes as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["vasnet", "dsn"])
def test_streaming_scorer_equals_resident_scoring(kind):
    from summarizer_amd.ingest import StreamingScorer
    from summarizer_amd.models.vasnet import VASNet
    from summarizer_amd.models.dsn import DSN
    dev = torch.device("cuda:0")
    D = 256
    torch.manual_seed(11)
    m = (VASNet(input_size=D) if kind == "vasnet" else DSN(D, 48, 1)).eval().to(dev)
    lens = [37, 1, 120, 64, 5, 300, 2, 90, 33, 150, 7]
    vids = [(f"video_{i}", R.features(T, 1, D, 300 + i)[:, 0, :].copy()) for i, T in enumerate(lens)]
    with torch.no_grad():
        want = {k: m.score_packed(torch.from_numpy(a).to(dev), [a.shape[0]]).cpu().numpy() for k, a in vids}
    for max_frames, depth in ((100000, 2), (256, 2), (130, 3), (64, 1)):       # one batch / several / a video longer than a slot
        got = list(StreamingScorer(m, max_frames=max_frames, depth=depth, pack_threads=3).score(vids))
        assert [k for k, _ in got] == [k for k, _ in vids]
        for k, s in got:
            np.testing.assert_array_equal(s, want[k], err_msg=f"{k} max_frames={max_frames} depth={depth}")


def test_streaming_scorer_rejects_bad_input():
    from summarizer_amd._lib import SumkError
    from summarizer_amd.ingest import StreamingScorer
    from summarizer_amd.models.vasnet import VASNet
    m = VASNet(input_size=64).eval().cuda()
    sc = StreamingScorer(m, max_frames=128)
    with pytest.raises(SumkError):
        list(sc.score([("a", np.zeros((4, 32), np.float32))]))          # wrong feature width
    with pytest.raises(SumkError):
        list(sc.score([("a", np.zeros((4, 64), np.float64))]))          # wrong dtype
    with pytest.raises(SumkError):
        StreamingScorer(VASNet(input_size=64, max_length=16).cuda())


def test_streaming_scorer_bf16_staging_equals_scoring_rounded_features():
    """stage_dtype="bf16" (on-the-fly bf16 conversion, half the PCIe bytes): lossy by construction -- the result
    must be EXACTLY the scoring of features rounded to bf16 beforehand, and stays close to the fp32 scoring (reported, loosely
    bounded: the parity gate of north_star applies to the fp32 staging, which is the default)."""
    from summarizer_amd.ingest import StreamingScorer
    from summarizer_amd.models.vasnet import VASNet
    dev = torch.device("cuda:0")
    D = 256
    torch.manual_seed(13)
    m = VASNet(input_size=D).eval().to(dev)
    lens = [37, 1, 120, 300, 2, 90]
    vids = [(f"video_{i}", R.features(T, 1, D, 500 + i)[:, 0, :].copy()) for i, T in enumerate(lens)]
    with torch.no_grad():
        rounded = {k: m.score_packed(torch.from_numpy(a).to(torch.bfloat16).float().to(dev), [a.shape[0]]).cpu().numpy() for k, a in vids}
        exact = {k: m.score_packed(torch.from_numpy(a).to(dev), [a.shape[0]]).cpu().numpy() for k, a in vids}
    worst = 0.0
    for max_frames in (100000, 130):
        got = list(StreamingScorer(m, max_frames=max_frames, depth=2, pack_threads=2, stage_dtype="bf16").score(vids))
        assert [k for k, _ in got] == [k for k, _ in vids]
        for k, s in got:
            np.testing.assert_array_equal(s, rounded[k], err_msg=k)
            worst = max(worst, float(np.abs(s - exact[k]).max()))
    print(f"bf16 staging: max |score - fp32-staged score| = {worst:.3e}")
    assert worst < 2e-2


def test_streaming_scorer_reads_an_npz_store(tmp_path):
    """score_store: an .npz feature store (entries "<video>/features", the layout DictDataset.save_npz writes) read one video at a
    time, in the store's order or in a requested key order; equals scoring the arrays directly."""
    from summarizer_amd.ingest import StreamingScorer, iter_store
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.utils.datasets import synthetic_dataset
    dev = torch.device("cuda:0")
    ds = synthetic_dataset(7, seed=21, D=128, t_range=(20, 90), n_users=3)
    path = str(tmp_path / "store.npz")
    ds.save_npz(path)
    torch.manual_seed(5)
    m = DSN(128, 32, 1).eval().to(dev)
    sc = StreamingScorer(m, max_frames=150, depth=2)
    keys = list(ds.keys())
    want = dict(sc.score((k, ds[k]["features"][...]) for k in keys))
    got = list(sc.score_store(path))
    assert sorted(k for k, _ in got) == sorted(keys)
    for k, s in got:
        np.testing.assert_array_equal(s, want[k])
    some = [keys[4], keys[0], keys[2]]
    got = list(sc.score_store(path, keys=some))
    assert [k for k, _ in got] == some
    for k, s in got:
        np.testing.assert_array_equal(s, want[k])
    got = list(sc.score_store(ds, keys=some))                     # an object with the h5py mapping protocol
    for k, s in got:
        np.testing.assert_array_equal(s, want[k])
    with pytest.raises(KeyError):
        list(iter_store(path, keys=["video_999"]))
