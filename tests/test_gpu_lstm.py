"""GPU parity: BiLSTM scorers (DSN, sLSTM) through the C ABI vs golden vectors from the real reference and
vs the numpy oracle.  Bar: 1e-4 on per-frame probabilities."""
import numpy as np
import pytest
import torch

import recipes as R
from conftest import load_golden, js

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _load(m, w, dev):
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    return m.eval().to(dev)


def test_lstm_small_goldens(dev):
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.models.sumgan import sLSTM
    g = load_golden("lstm_small")
    for name, mk in [("dsn_small", lambda: DSN(64, 16, 1)), ("dsn_small_2l", lambda: DSN(64, 16, 2)), ("slstm_small", lambda: sLSTM(64, 32, 2))]:
        w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith(f"{name}/w/")}
        m = _load(mk(), w, dev)
        for c in sorted(k.split("/")[-1] for k in g.files if k.startswith(f"{name}/x/")):
            with torch.no_grad():
                y = m(torch.from_numpy(g[f"{name}/x/{c}"].copy()).to(dev)).cpu().numpy()
            np.testing.assert_allclose(y, g[f"{name}/y/{c}"], atol=TOL, rtol=0, err_msg=f"{name} {c}")


def test_lstm_hidden_states_vs_golden(dev):
    from summarizer_amd import kernels
    from summarizer_amd.models.dsn import DSN
    g = load_golden("lstm_small")
    w = {k.split("/w/")[1]: g[k] for k in g.files if k.startswith("dsn_small/w/")}
    m = _load(DSN(64, 16, 1), w, dev)
    x = torch.from_numpy(g["dsn_small/x/T37B1"][:, 0, :].copy()).to(dev)
    sb = kernels.SeqBatch.get([37], dev)
    h, _ = kernels.bilstm_layer_forward(x, sb, dict(m.named_parameters()), "rnn.", 0, 16)
    np.testing.assert_allclose(h.detach().cpu().numpy(), g["dsn_small/h/T37B1"][:, 0, :], atol=2e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_lstm_full_size_goldens(dev, precision):
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.models.sumgan import sLSTM
    g = load_golden("lstm_full")
    for ci in range(3):
        cfg = js(g[f"c{ci}/cfg"])
        if cfg["kind"] == "dsn":
            w = R.lstm_weights("rnn.", cfg["D"], cfg["H"], cfg["L"], cfg["wseed"], "out.0."); m = DSN(cfg["D"], cfg["H"], cfg["L"])
        else:
            w = R.lstm_weights("lstm.", cfg["D"], cfg["H"], cfg["L"], cfg["wseed"], "out."); m = sLSTM(cfg["D"], cfg["H"], cfg["L"])
        assert R.digest(w) == cfg["wdigest"]
        m = _load(m, w, dev)
        m.precision = precision     # bf16x3: input projections (and the H = 1024 recurrent product) in split-bf16 arithmetic
        with torch.no_grad():
            y = m(torch.from_numpy(R.features(cfg["T"], 1, cfg["D"], cfg["xseed"])).to(dev)).cpu().numpy()
        np.testing.assert_allclose(y, g[f"c{ci}/y"], atol=TOL, rtol=0, err_msg=str(cfg))
        print(precision, cfg["kind"], cfg["T"], "max |d| vs reference:", float(np.abs(y - g[f"c{ci}/y"]).max()))


def test_dsn_packed_ragged_batch_vs_oracle(dev):
    from oracle import lstm_np
    from summarizer_amd.models.dsn import DSN
    D, H = 128, 40          # H not a multiple of 8/32: exercises the k-chunk and unit-block tails
    w = R.lstm_weights("rnn.", D, H, 1, 77, "out.0.")
    m = _load(DSN(D, H, 1), w, dev)
    lens = [1, 2, 33, 64, 5, 100] + [3] * 30          # > 32 videos: two M tiles in the step kernel
    xs = [R.features(T, 1, D, 500 + i) - 0.2 for i, T in enumerate(lens)]
    with torch.no_grad():
        s = m.score_packed(torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev), lens).cpu().numpy()
    off = np.concatenate([[0], np.cumsum(lens)])
    for i, x in enumerate(xs):
        ref = lstm_np.dsn_forward(x, w)[:, 0, 0]
        np.testing.assert_allclose(s[off[i]:off[i + 1]], ref, atol=TOL, rtol=0, err_msg=f"video {i} T={lens[i]}")


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_wide_recurrence_ragged_batch_vs_oracle(dev, precision):
    """256 < H <= 1024 runs the two-team register-resident recurrence (lstm_wide_kernel): H = 320 is not a multiple of
    128 (3 units per member, last member short), 70 videos = two work items per direction, lengths 1 ... 90."""
    from oracle import lstm_np
    from summarizer_amd.models.dsn import DSN
    D, H = 64, 320
    w = R.lstm_weights("rnn.", D, H, 1, 78, "out.0.")
    m = _load(DSN(D, H, 1), w, dev)
    m.precision = precision
    lens = [1, 2, 33, 64, 5, 90] + [3, 7] * 32
    xs = [R.features(T, 1, D, 900 + i) - 0.2 for i, T in enumerate(lens)]
    with torch.no_grad():
        s = m.score_packed(torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev), lens).cpu().numpy()
    off = np.concatenate([[0], np.cumsum(lens)])
    for i, x in enumerate(xs):
        ref = lstm_np.dsn_forward(x, w)[:, 0, 0]
        np.testing.assert_allclose(s[off[i]:off[i + 1]], ref, atol=TOL, rtol=0, err_msg=f"video {i} T={lens[i]}")


def test_slstm_batch_properties_full_size(dev):
    """sLSTM (2 layers, H = 1024) at BASELINE size: a video's scores do not depend on what it is batched with, bit for bit."""
    from summarizer_amd.models.sumgan import sLSTM
    torch.manual_seed(4)
    m = sLSTM().eval().to(dev)
    lens = [int(np.ceil(v)) for v in np.random.default_rng(2).uniform(150, 320, 6)]
    xs = [torch.from_numpy(R.features(T, 1, 1024, 800 + i)[:, 0, :]).to(dev) for i, T in enumerate(lens)]
    with torch.no_grad():
        a = m.score_packed(torch.cat(xs), lens)
        b = m.score_packed(torch.cat(xs[::-1]), lens[::-1])
        single = m(xs[2].unsqueeze(1))[:, 0, 0]
    off = np.concatenate([[0], np.cumsum(lens)]); offr = np.concatenate([[0], np.cumsum(lens[::-1])])
    for i in range(len(lens)):
        j = len(lens) - 1 - i
        assert torch.equal(b[offr[j]:offr[j + 1]], a[off[i]:off[i + 1]])
    assert torch.equal(single, a[off[2]:off[3]])
    assert bool(((a > 0) & (a < 1)).all())


def test_dsn_batch_properties_full_size(dev):
    from summarizer_amd.models.dsn import DSN
    torch.manual_seed(3)
    m = DSN().eval().to(dev)
    lens = [int(np.ceil(v)) for v in np.random.default_rng(1).uniform(150, 320, 10)]
    xs = [torch.from_numpy(R.features(T, 1, 1024, 700 + i)[:, 0, :]).to(dev) for i, T in enumerate(lens)]
    with torch.no_grad():
        a = m.score_packed(torch.cat(xs), lens)
        b = m.score_packed(torch.cat(xs[::-1]), lens[::-1])
        singles = [m(x.unsqueeze(1))[:, 0, 0] for x in xs]
    off = np.concatenate([[0], np.cumsum(lens)]); offr = np.concatenate([[0], np.cumsum(lens[::-1])])
    for i in range(len(lens)):
        j = len(lens) - 1 - i
        assert torch.equal(b[offr[j]:offr[j + 1]], a[off[i]:off[i + 1]])
        assert torch.equal(singles[i], a[off[i]:off[i + 1]])
    assert bool(((a > 0) & (a < 1)).all())


@pytest.mark.parametrize("kind,T", [("dsn", 3000), ("slstm", 400)])
def test_bilstm_time_reversal_symmetry_full_size(dev, kind, T):
    """Size-independent property at BASELINE sizes (D = 1024; DSN H = 256 at T = 3000, sLSTM 2 x H = 1024 at T = 400):
    a bidirectional LSTM run on the time-reversed video with its forward / reverse weights swapped produces the
    time-reversed hidden sequence with the two halves swapped -- so, with the head's halves swapped too, the reversed scores."""
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.models.sumgan import sLSTM
    torch.manual_seed(6)
    m = (DSN() if kind == "dsn" else sLSTM()).eval().to(dev)
    pre, hw = ("rnn.", "out.0.weight") if kind == "dsn" else ("lstm.", "out.weight")
    H = m.hidden_size
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sw = {}
    for k, v in sd.items():
        if k.startswith(pre):
            other = k[:-8] if k.endswith("_reverse") else k + "_reverse"
            v = sd[other].clone()
            layer = int(k.split("_l")[1].split("_")[0])
            if "weight_ih" in k and layer > 0:                       # deeper layers see [fwd || rev] inputs: swap the halves
                v = torch.cat([v[:, H:], v[:, :H]], dim=1)
        elif k == hw:
            v = torch.cat([v[:, H:], v[:, :H]], dim=1)
        sw[k] = v
    m2 = (DSN() if kind == "dsn" else sLSTM()).eval().to(dev)
    m2.load_state_dict(sw)
    x = torch.from_numpy(R.features(T, 1, 1024, 4242)[:, 0, :]).to(dev)
    with torch.no_grad():
        a = m.score_packed(x, [T])
        b = m2.score_packed(torch.flip(x, (0,)).contiguous(), [T])
    np.testing.assert_allclose(torch.flip(b, (0,)).cpu().numpy(), a.cpu().numpy(), atol=1e-6, rtol=0)


def test_persistent_recurrences_are_bitwise_repeatable_under_foreign_traffic():
    """The hand-off protocol of the persistent recurrence kernels (forward and BPTT, H = 256 and H = 1024) under repetition: the
    same training step from identical inputs must give bit-identical scores and gradients every time, also while another buffer
    is being streamed through the caches between repetitions -- a stale or torn read in a hand-off would show as a difference
    (scripts/probes/wide_bptt_soak.py, 2 x 6 repetitions per model here; 2 x 40 were run for DESIGN.md)."""
    import os, subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probes", "wide_bptt_soak.py"), "6"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 differed" in r.stdout


_HANDOFF_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import recipes as R
from summarizer_amd.models.dsn import DSN
D, H, lens = 128, int(sys.argv[2]), [int(v) for v in sys.argv[3].split(",")]
w = R.lstm_weights("rnn.", D, H, 1, 21, "out.0.")
m = DSN(D, H, 1); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to("cuda:0")
xp = torch.from_numpy(np.concatenate([(R.features(T, 1, D, 80 + i) - 0.2)[:, 0, :] for i, T in enumerate(lens)])).to("cuda:0").requires_grad_(True)
s = m.score_packed(xp, lens)
cw = torch.from_numpy(np.random.default_rng(4).standard_normal(sum(lens)).astype(np.float32)).to("cuda:0")
(s * cw).sum().backward()
np.savez(sys.argv[1], scores=s.detach().cpu().numpy(), gx=xp.grad.cpu().numpy(), **{k: p.grad.cpu().numpy() for k, p in m.named_parameters()})
'''


@pytest.mark.parametrize("H,lens", [(32, [20, 10, 20, 1, 20, 10, 20, 7, 20]), (40, [50, 1, 33, 7] + [4] * 30), (256, [37, 64, 12] * 14)])
def test_flag_in_data_handoff_equals_counter_handoff(tmp_path, H, lens):
    """The persistent recurrences hand h_t (forward) and the partial products of dh (BPTT) from member to member either through a step
    counter (stores, vmcnt(0) drain, barrier, atomic add; consumers poll the counter) or as 8-byte {value, step tag} packets whose own
    loads are the poll, on 32-row or -- groups of <= 16 videos -- 16-row MFMAs.  Defaults: forward flag-in-data, BPTT counter (its
    32-way reduce-scatter measured slower with packets), 16-row MFMAs in both.  Reference here = the round-2 path (counter hand-off,
    32-row MFMAs: SUMK_LSTM_LL=0 SUMK_LSTM_M16=0); against it the default and the all-packets variant (SUMK_LSTM_LL_BWD=1) must agree
    to fp32 re-association of the 16- vs 32-row k order (1e-6 of the largest entry) in scores and every gradient -- on ragged groups
    of SEVERAL videos (rows > 0 of a group's exchange block: a one-video group cannot see a row mix-up), with one-frame videos, H not
    a multiple of 32, and groups of more than 16 videos."""
    import os, subprocess, sys
    out = {}
    for tag, env in (("ref", {"SUMK_LSTM_LL": "0", "SUMK_LSTM_M16": "0"}), ("default", {}), ("packets", {"SUMK_LSTM_LL_BWD": "1"})):
        f = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", _HANDOFF_CHILD, str(f), str(H), ",".join(map(str, lens))], env=dict(os.environ, **env),
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = dict(np.load(f))
    for tag in ("default", "packets"):
        for k in out[tag]:
            a, b = out[tag][k], out["ref"][k]
            assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max() + 1e-12, (tag, k, float(np.abs(a - b).max()), float(np.abs(b).max()))


def test_dsn_backward_with_tail_event_equals_plain_backward(dev):
    """sumk_lstm_layer_grads::tail_ready_event (the data-parallel overlap of DSNTrainer: biases and the reverse direction first, an event, then the
    forward direction): the same gradients as the plain order -- the two directions' dW_ih run as two split-K launches instead of one, so equal to
    summation order (2e-5 relative), and the event has fired by the time the call's stream is idle."""
    from summarizer_amd.models.dsn import DSN
    torch.manual_seed(5)
    D = 1024
    lens = [int(t) for t in np.random.default_rng(4).integers(150, 321, size=6)]
    x = torch.from_numpy(np.concatenate([R.features(T, 1, D, 50 + i)[:, 0, :] for i, T in enumerate(lens)])).to(dev)
    m = DSN(input_size=D).to(dev).train()
    wgt = torch.linspace(-1, 1, x.shape[0], device=dev)

    def grads(event):
        m.zero_grad(set_to_none=True)
        m.tail_grads_ready_event = event
        (m.score_packed(x, lens) * wgt).sum().backward()
        m.tail_grads_ready_event = None
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    plain = grads(None)
    ev = torch.cuda.Event()
    with_ev = grads(ev)
    torch.cuda.synchronize()
    assert ev.query()
    for k in plain:
        a, b = plain[k], with_ev[k]
        rel = float((a - b).norm() / a.norm().clamp_min(1e-30))
        assert rel < 2e-5, (k, rel)
    for k in ("rnn.weight_hh_l0", "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0", "rnn.bias_hh_l0_reverse", "out.0.weight"):
        assert torch.equal(plain[k], with_ev[k]), k          # the launches that did not change shape are bit-identical


_WIDE_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import recipes as R
from summarizer_amd.models.dsn import DSN
D, H, lens = 64, int(sys.argv[2]), [int(v) for v in sys.argv[3].split(",")]
w = R.lstm_weights("rnn.", D, H, 1, 31, "out.0.")
m = DSN(D, H, 1); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to("cuda:0").eval()
x = torch.from_numpy(np.concatenate([(R.features(T, 1, D, 60 + i) - 0.2)[:, 0, :] for i, T in enumerate(lens)])).to("cuda:0")
out = {}
for prec in ("fp32", "bf16x3", "bf16x6"):
    m.precision = prec
    with torch.no_grad():
        out[prec] = m.score_packed(x, lens).cpu().numpy()
m.precision = "fp32"; m.train()
xg = x.clone().requires_grad_(True)
s = m.score_packed(xg, lens)
(s * torch.linspace(-1, 1, s.numel(), device=s.device)).sum().backward()
out["train_scores"] = s.detach().cpu().numpy(); out["gx"] = xg.grad.cpu().numpy()
for k, p in m.named_parameters(): out["g_" + k] = p.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
'''


@pytest.mark.parametrize("H,lens", [(1024, [70, 33, 1, 70, 12] * 8 + [5, 64, 64]), (320, [1, 2, 33, 64, 5, 90] + [3, 7] * 32), (512, [40] * 33 + [9, 17])])
def test_wide_recurrence_forms_agree(tmp_path, H, lens):
    """lstm_wide2_kernel (round 6: exchange buffer laid out for the consumers, rows sorted by length with the second MFMA tile dropped once
    fewer than 33 videos run, sharded step counter, stores behind the signal) against lstm_wide_kernel (SUMK_LSTM_WIDE2=0): the k order of
    every output element is the same, so exact fp32 and bf16x3 must agree BIT FOR BIT -- inference scores, training-mode scores and every
    gradient (the BPTT reads the forward's saves) -- on ragged groups (43 and 70 videos: one and two work items per direction, tile
    boundaries at 32 / 33 videos, one-frame videos, ties in length), at H = 1024 (packed 16-byte publish), H = 320 (3 units per member,
    scalar publish, k tail) and H = 512 (4 units per member).  The fp32-grade bf16x6 recurrence exists in the new form only: within 2e-6
    of exact fp32 (the old form ran the recurrent product of that mode in fp32)."""
    import os, subprocess, sys
    out = {}
    for tag, env in (("old", {"SUMK_LSTM_WIDE2": "0"}), ("new", {})):
        f = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", _WIDE_CHILD, str(f), str(H), ",".join(map(str, lens))], env=dict(os.environ, **env),
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = dict(np.load(f))
    for k in out["new"]:
        if k == "bf16x6":
            continue
        assert np.array_equal(out["new"][k], out["old"][k]), (k, float(np.abs(out["new"][k] - out["old"][k]).max()))
    assert np.abs(out["new"]["bf16x6"] - out["new"]["fp32"]).max() < 2e-6
    assert np.abs(out["new"]["bf16x6"] - out["new"]["fp32"]).max() > 0 or H < 1024      # (a different arithmetic did run)


_PROJ_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import recipes as R
from summarizer_amd.models.dsn import DSN
D, H = 1024, int(sys.argv[2])
lens = [int(v) for v in sys.argv[3].split(",")]
w = R.lstm_weights("rnn.", D, H, 1, 41, "out.0.")
m = DSN(D, H, 1); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to("cuda:0").eval()
x = torch.from_numpy(np.concatenate([(R.features(T, 1, D, 90 + i) - 0.2)[:, 0, :] for i, T in enumerate(lens)])).to("cuda:0")
out = {}
for prec in ("fp32", "bf16x6", "bf16x3"):
    m.precision = prec
    with torch.no_grad():
        out[prec] = m.score_packed(x, lens).cpu().numpy()
        out[prec + "_again"] = m.score_packed(x, lens).cpu().numpy()
np.savez(sys.argv[1], **out)
'''


@pytest.mark.parametrize("H,lens", [(256, [300, 1, 2, 177, 320, 33] + [64, 150, 7] * 14), (96, [200, 3, 150, 150, 1, 90, 320, 17] + [29] * 9)])
def test_dsn_projection_inside_the_recurrence(tmp_path, H, lens):
    """Round 6: in the split-bf16 inference modes (operand planes at hand, groups of <= 16 videos, In = 1024) the input projection of the
    H <= 256 BiLSTM runs INSIDE the persistent recurrence (lstm_persist_proj_kernel: the step's x rows times the member's W_ih fragments
    start the accumulators of the recurrent product; G is never formed).  Against the plane GEMM in front of the recurrence
    (SUMK_LSTM_PROJ=0) the scores agree to summation order -- 2e-6 at three planes, 1e-5 at two -- on ragged batches with one- and
    two-frame videos, 48 videos (groups of 12) and 17 (groups of 5; H = 96: three units per member, k tail of the recurrent product), and
    every video stays within 1e-4 of the numpy oracle (tests/golden/recipes + oracle/lstm_np.py); a second call is bit-identical."""
    import os, subprocess, sys
    from oracle import lstm_np
    out = {}
    for tag, env in (("gemm", {"SUMK_LSTM_PROJ": "0"}), ("fused", {"SUMK_LSTM_PROJ": "1"})):      # (opt-in since the round's last pass: the plane GEMM in front measured faster)
        f = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", _PROJ_CHILD, str(f), str(H), ",".join(map(str, lens))], env=dict(os.environ, **env),
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = dict(np.load(f))
    assert np.array_equal(out["fused"]["fp32"], out["gemm"]["fp32"])          # exact fp32 does not take the fused path
    for prec, tol in (("bf16x6", 2e-6), ("bf16x3", 1e-5)):
        a, b = out["fused"][prec], out["gemm"][prec]
        assert np.array_equal(a, out["fused"][prec + "_again"])
        assert np.abs(a - b).max() < tol, (prec, float(np.abs(a - b).max()))
        assert not np.array_equal(a, b) or H != 256                            # (a different kernel did run at the headline shape)
    w = R.lstm_weights("rnn.", 1024, H, 1, 41, "out.0.")
    off = np.concatenate([[0], np.cumsum(lens)])
    for i in (0, 1, 2, 3, len(lens) - 1):
        ref = lstm_np.dsn_forward(R.features(lens[i], 1, 1024, 90 + i) - 0.2, w)[:, 0, 0]
        for prec in ("bf16x6", "bf16x3"):
            np.testing.assert_allclose(out["fused"][prec][off[i]:off[i + 1]], ref, atol=TOL, rtol=0, err_msg=f"{prec} video {i} T={lens[i]}")


@pytest.mark.parametrize("H,lens", [(256, [37, 64, 12] * 14), (256, [5, 9, 1, 30] * 35), (40, [50, 1, 33, 7] + [4] * 30), (384, [20, 7, 33, 12, 5, 40, 3, 18, 25, 9, 1] * 7)])
def test_bptt_round6_exchange_equals_round5_exchange(tmp_path, H, lens):
    """Round 6, persistent BPTT (counter hand-off): the exchange of the members' partial products laid out for its READERS
    ([reader][producer][video][8 columns]: the 32 partials a cell-update thread sums come out of one contiguous block) and the step counter
    as four shards on lines of their own -- against the round-5 forms (SUMK_LSTM_BWD_R6=0: producer-major rows, one counter word).  Same
    products, same fixed summation order: scores and EVERY gradient bit for bit, on 42 videos (groups of 11: 16-row MFMAs, 16-byte
    publish), 140 videos (groups of 32: 32-row MFMAs, scalar publish), H = 40 (members of 2 units: the round-5 layout stays, the
    sharded counter does not) and H = 384 with 77 videos (lstm_wide_bwd_kernel: two groups per direction, the sharded counter alone)."""
    import os, subprocess, sys
    out = {}
    for tag, env in (("r5", {"SUMK_LSTM_BWD_R6": "0"}), ("r6", {})):
        f = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", _HANDOFF_CHILD, str(f), str(H), ",".join(map(str, lens))], env=dict(os.environ, **env),
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = dict(np.load(f))
    for k in out["r6"]:
        assert np.array_equal(out["r6"][k], out["r5"][k]), (k, float(np.abs(out["r6"][k] - out["r5"][k]).max()))
