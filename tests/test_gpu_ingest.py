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
