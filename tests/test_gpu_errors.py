"""GPU: error behaviour of the C ABI -- bad shapes, short workspaces and null pointers come back as negative status codes
with a message (never a crash, never a silent fallback), and edge-case inputs (single-frame videos) are handled."""
import ctypes as C
import numpy as np
import pytest
import torch

import recipes as R

pytestmark = pytest.mark.gpu


def test_status_codes_and_messages():
    from summarizer_amd import _lib, kernels
    from summarizer_amd._lib import SumkError
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = torch.zeros(8, 8, device=dev)
    assert lib.sumk_gemm_nt(a.data_ptr(), a.data_ptr(), a.data_ptr(), 8, 8, 6, st) == -1        # K not a multiple of 4
    assert b"multiples of 4" in lib.sumk_last_error()
    assert lib.sumk_gemm_nt(None, a.data_ptr(), a.data_ptr(), 8, 8, 8, st) == -1
    # bf16-source GEMM: ineligible shapes and bad split-K workspaces are refused with a message, nothing is launched
    a16 = torch.zeros(128, 128, dtype=torch.bfloat16, device=dev); c32 = torch.zeros(128, 128, device=dev)
    assert lib.sumk_gemm_bf16src(0, a16.data_ptr(), a16.data_ptr(), c32.data_ptr(), 128, 128, 96, None, 0, st) == -1     # K % 64 for K-contiguous operands
    assert b"not eligible" in lib.sumk_last_error()
    assert lib.sumk_gemm_bf16src(2, a16.data_ptr(), a16.data_ptr(), c32.data_ptr(), 100, 128, 128, None, 0, st) == -1   # M % 8 for an M-contiguous operand
    assert lib.sumk_gemm_bf16src(3, a16.data_ptr(), a16.data_ptr(), c32.data_ptr(), 128, 128, 128, None, 0, st) == -1   # no such layout
    wsb = torch.zeros(8192 + 128 * 128 * 4, dtype=torch.uint8, device=dev)
    assert lib.sumk_gemm_bf16src(0, a16.data_ptr(), a16.data_ptr(), c32.data_ptr(), 128, 128, 128, wsb.data_ptr(), wsb.numel(), st) == -1   # split-K is the TN form
    assert b"TN form" in lib.sumk_last_error()
    assert lib.sumk_gemm_bf16src(2, a16.data_ptr(), a16.data_ptr(), c32.data_ptr(), 128, 128, 128, wsb.data_ptr(), 1024, st) == -1        # workspace too small
    assert b"workspace" in lib.sumk_last_error()
    assert lib.sumk_gemm_bf16src(2, a16.data_ptr(), a16.data_ptr(), c32.data_ptr(), 128, 128, 128, wsb.data_ptr(), wsb.numel(), st) == 0
    off = np.array([0, 5, 5], dtype=np.int32)                                                    # empty video
    assert lib.sumk_vasnet_workspace_bytes(64, 2, _lib.host_i32(off), 0) == 0
    assert b"has 0 frames" in lib.sumk_last_error()
    assert lib.sumk_vasnet_workspace_bytes(62, 1, _lib.host_i32(np.array([0, 5], dtype=np.int32)), 0) == 0   # D % 4 != 0
    # workspace too small -> SUMK_ERR_WORKSPACE (-2)
    from summarizer_amd.models.vasnet import VASNet
    m = VASNet(input_size=64).eval().to(dev)
    sb = kernels.SeqBatch.get([9], dev)
    w, o = kernels._vasnet_structs(dict(m.named_parameters()), dict(scale=0.125, eps=1e-6))
    x = torch.zeros(9, 64, device=dev); out = torch.zeros(9, device=dev); ws = torch.zeros(64, dtype=torch.uint8, device=dev)
    rc = lib.sumk_vasnet_forward(x.data_ptr(), 64, 1, sb.off_host_p, sb.off_dev_p, C.byref(w), C.byref(o), None, None,
                                 out.data_ptr(), ws.data_ptr(), ws.numel(), 0, st)
    assert rc == -2 and b"workspace" in lib.sumk_last_error()
    # dropout outside training mode is refused
    o.dropout_p = 0.5
    big = torch.zeros(lib.sumk_vasnet_workspace_bytes(64, 1, sb.off_host_p, 0), dtype=torch.uint8, device=dev)
    rc = lib.sumk_vasnet_forward(x.data_ptr(), 64, 1, sb.off_host_p, sb.off_dev_p, C.byref(w), C.byref(o), None, None,
                                 out.data_ptr(), big.data_ptr(), big.numel(), 0, st)
    assert rc == -1 and b"dropout" in lib.sumk_last_error()
    # python layer: wrong dtype / shape / device raise SumkError
    with pytest.raises(SumkError):
        kernels.vasnet_forward_packed(torch.zeros(9, 64, device=dev, dtype=torch.float64), sb, dict(m.named_parameters()), dict(scale=1, eps=1e-6))
    with pytest.raises(SumkError):
        kernels.vasnet_forward_packed(torch.zeros(8, 64, device=dev), sb, dict(m.named_parameters()), dict(scale=1, eps=1e-6))
    with pytest.raises(SumkError):
        kernels.SeqBatch([3, 0], dev)
    # loss-glue entries: any number of episodes runs (the reference takes any num_episodes, dsn.py:53; 17 > one register chunk of 16);
    # a non-positive count / null pointers are refused
    pr = torch.rand(9, device=dev); ac = torch.zeros(17, 9, device=dev); rw = torch.zeros(17, 1, device=dev); bs = torch.zeros(1, device=dev)
    lv = torch.zeros(1, device=dev); mp = torch.zeros(1, device=dev)
    rc = lib.sumk_dsn_policy_loss_forward(pr.data_ptr(), ac.data_ptr(), rw.data_ptr(), bs.data_ptr(), 1, 9, sb.off_dev_p, 17, 0.01, 0.5,
                                          lv.data_ptr(), mp.data_ptr(), st)
    assert rc == 0
    rc = lib.sumk_dsn_policy_loss_forward(pr.data_ptr(), ac.data_ptr(), rw.data_ptr(), bs.data_ptr(), 1, 9, sb.off_dev_p, 0, 0.01, 0.5,
                                          lv.data_ptr(), mp.data_ptr(), st)
    assert rc == -1 and b"episodes=0" in lib.sumk_last_error()
    assert lib.sumk_segment_mse_forward(pr.data_ptr(), None, 1, sb.off_dev_p, lv.data_ptr(), st) == -1
    # round-4 entries: the fused step loss needs its ticket word, the two-stage device tail its scratch; zero videos is a no-op
    assert lib.sumk_segment_mse_mean_forward(pr.data_ptr(), pr.data_ptr(), 1, sb.off_dev_p, 1.0, lv.data_ptr(), mp.data_ptr(), None, st) == -1
    assert lib.sumk_segment_mse_mean_backward(pr.data_ptr(), pr.data_ptr(), None, 1.0, 1, sb.off_dev_p, pr.data_ptr(), st) == -1
    assert lib.sumk_eval_device_segments(pr.data_ptr(), None, 1, pr.data_ptr(), pr.data_ptr(), st) == -1
    assert lib.sumk_eval_device_spearman(pr.data_ptr(), pr.data_ptr(), 1, None, pr.data_ptr(), st) == -1
    assert lib.sumk_eval_device_segments(None, None, 0, None, None, st) == 0 and lib.sumk_eval_device_spearman(None, None, 0, None, None, st) == 0
    assert lib.sumk_eval_device_spearman_scratch_bytes(50) == 50 * 8 * 33 * 8 and lib.sumk_eval_device_spearman_scratch_bytes(-3) == 0
    assert lib.sumk_adam_step_dev(pr.data_ptr(), pr.data_ptr(), pr.data_ptr(), pr.data_ptr(), 9, 1e-3, 0.9, 0.999, 1e-8, 0.0, None, 1.0,
                                  None, 0.0, st) == -1
    from summarizer_amd.models.dsn import DSN
    assert isinstance(DSN(cell="gru").rnn, torch.nn.GRU)         # the optional cell exists (tests/test_gpu_gru.py)
    with pytest.raises(AssertionError):
        DSN(cell="rnn")                     # dsn.py:21


def test_single_frame_videos_everywhere():
    from oracle import vasnet_np, lstm_np
    from summarizer_amd.models.vasnet import VASNet
    from summarizer_amd.models.dsn import DSN
    dev = torch.device("cuda:0")
    D = 64
    wv = R.vasnet_weights(D, 1); wl = R.lstm_weights("rnn.", D, 16, 1, 2, "out.0.")
    mv = VASNet(input_size=D).eval(); mv.load_state_dict({k: torch.from_numpy(v) for k, v in wv.items()}); mv = mv.to(dev)
    ml = DSN(D, 16, 1).eval(); ml.load_state_dict({k: torch.from_numpy(v) for k, v in wl.items()}); ml = ml.to(dev)
    xs = [R.features(1, 1, D, 10 + i) for i in range(40)]                # 40 one-frame videos in one packed batch
    xp = torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to(dev)
    with torch.no_grad():
        sv = mv.score_packed(xp, [1] * 40).cpu().numpy(); sl = ml.score_packed(xp, [1] * 40).cpu().numpy()
    for i, x in enumerate(xs):
        np.testing.assert_allclose(sv[i], vasnet_np.vasnet_forward(x, wv)[0, 0, 0], atol=1e-4)
        np.testing.assert_allclose(sl[i], lstm_np.dsn_forward(x, wl)[0, 0, 0], atol=1e-4)


def test_integration_md_binding_stub_runs():
    """The ctypes stub printed in INTEGRATION.md section 2 is executed verbatim against libsumk.so: it must score a batch
    exactly like the Python mirror does (keeps the document honest when the ABI moves)."""
    import os, re
    import numpy as np
    import torch
    from conftest import ROOT
    from summarizer_amd.models.vasnet import VASNet
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "def vasnet_scores" in b)
    cwd = os.getcwd()
    os.chdir(ROOT)                      # the stub loads "summarizer_amd/libsumk.so" relative to the repo root
    try:
        ns = {}
        exec(stub, ns)
        torch.manual_seed(2)
        m = VASNet().cuda().eval()
        feats = [torch.rand(T, 1024, device="cuda") * 0.5 for T in (40, 7, 130)]
        got = ns["vasnet_scores"](m, feats)
        with torch.no_grad():
            want = m.score_packed(torch.cat(feats), [f.shape[0] for f in feats])
        assert torch.equal(torch.cat(list(got)), want)
    finally:
        os.chdir(cwd)


def test_rccl_entry_point_single_rank():
    """`sumk_allreduce_flat` (the data-parallel exchange behind the C ABI): bootstrap a one-rank communicator on the
    test box's single GPU and reduce an fp32 and a bf16 bucket in place on the current stream -- a one-rank SUM is the identity,
    which checks the dlopen of RCCL, the by-value unique id, the dtype / op codes and the stream argument.  (Two ranks need two
    GPUs: RCCL refuses duplicate devices.)"""
    import ctypes as C
    import torch
    from summarizer_amd import _lib
    lib = _lib.load()
    ident = (C.c_uint8 * 128)()
    _lib.check(lib.sumk_comm_unique_id(ident), "unique id")
    assert any(ident)
    comm = C.c_void_p()
    _lib.check(lib.sumk_comm_init(ident, 0, 1, C.byref(comm)), "init")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = torch.randn(1 << 20, device="cuda"); a0 = a.clone()
    _lib.check(lib.sumk_allreduce_flat(comm, C.c_void_p(a.data_ptr()), a.numel(), 0, st), "allreduce f32")
    b = torch.randn(4099, device="cuda").to(torch.bfloat16); b0 = b.clone()
    _lib.check(lib.sumk_allreduce_flat(comm, C.c_void_p(b.data_ptr()), b.numel(), 1, st), "allreduce bf16")
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(b, b0)
    assert lib.sumk_allreduce_flat(comm, C.c_void_p(a.data_ptr()), a.numel(), 7, st) != 0 and b"dtype" in lib.sumk_last_error()
    _lib.check(lib.sumk_comm_destroy(comm), "destroy")


def test_flat_adam_through_the_rccl_entry_point(monkeypatch):
    """SUMK_RCCL_DIRECT=1: FlatAdam's gradient exchange goes through RcclDirect (here a one-rank group over gloo for the
    bootstrap): the step must equal the plain single-process step."""
    import os, socket
    import torch
    import torch.distributed as dist
    from summarizer_amd import training
    from summarizer_amd.training import FlatAdam
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    monkeypatch.setenv("SUMK_RCCL_DIRECT", "1")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        monkeypatch.setattr(training, "dist_info", lambda: (0, 2))        # pretend a 2-rank job so the collective path is taken
        torch.manual_seed(0)
        p = torch.nn.Parameter(torch.randn(1000, device="cuda")); q = torch.nn.Parameter(p.detach().clone())
        g = torch.randn(1000, device="cuda")
        a = FlatAdam([p], lr=1e-2); b = FlatAdam([q], lr=1e-2)
        a.flat_grad[:1000] = g; b.flat_grad[:1000] = g
        monkeypatch.setattr(training.RcclDirect, "__init__", _one_rank_init)
        scale = a.all_reduce_grads()                    # one-rank RCCL sum == identity; scale = 1/2 of the pretended world
        assert scale == 0.5
        a.step(grad_scale=1.0); b.step(grad_scale=1.0)
        torch.testing.assert_close(p.detach(), q.detach(), rtol=0, atol=0)
        training.RcclDirect.get().close()
    finally:
        dist.destroy_process_group()


def _one_rank_init(self):
    import ctypes as C
    from summarizer_amd import _lib
    self._lib, self._C = _lib.load(), C
    ident = (C.c_uint8 * 128)()
    _lib.check(self._lib.sumk_comm_unique_id(ident), "sumk_comm_unique_id")
    comm = C.c_void_p()
    _lib.check(self._lib.sumk_comm_init(ident, 0, 1, C.byref(comm)), "sumk_comm_init")
    self.comm, self.world = comm, 1


def test_scoring_call_is_hip_graph_capturable(monkeypatch):
    """INTEGRATION.md states that every entry point only enqueues on the caller's stream (no hidden allocation, no host sync):
    a packed scoring call is captured into a HIP graph (torch.cuda.CUDAGraph: the ABI receives the capturing stream) and the
    replays reproduce the eager scores bit for bit, also after the input buffer was refilled in place."""
    from summarizer_amd import kernels
    from summarizer_amd.models.vasnet import VASNet
    from summarizer_amd.models.dsn import DSN
    monkeypatch.setattr(kernels, "CHECK_LSTM", False)    # the test-suite's per-layer health check synchronises: not inside a capture
    dev = torch.device("cuda:0")
    D, lens = 256, [70, 1, 33, 129, 64]
    for make in (lambda: VASNet(input_size=D), lambda: DSN(input_size=D, hidden_size=32)):
        torch.manual_seed(3)
        m = make().to(dev).eval()
        xs = [torch.from_numpy(np.concatenate([R.features(T, 1, D, 10 * k + i)[:, 0, :] for i, T in enumerate(lens)])).to(dev) for k in range(2)]
        with torch.no_grad():
            eager = [m.score_packed(x, lens).clone() for x in xs]
            static_x = xs[0].clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m.score_packed(static_x, lens)                     # warm-up on the side stream (workspace cache, table build)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = m.score_packed(static_x, lens)
            for k in (0, 1, 0):
                static_x.copy_(xs[k])
                graph.replay()
                torch.cuda.synchronize()
                assert torch.equal(static_out, eager[k]), (type(m).__name__, k)


def test_training_step_is_hip_graph_capturable_and_replays_advance_adam(monkeypatch):
    """A whole training step -- forward, loss, autograd backward through the libsumk Functions, grad-norm clip and Adam -- has no
    host synchronisation (the optimiser's step counter and clip coefficient live on the device: sumk_adam_step_dev), so it can be
    captured into a HIP graph: one eager step + three replays must leave exactly the weights of four eager steps."""
    from summarizer_amd import kernels
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.training import FlatAdam
    monkeypatch.setattr(kernels, "CHECK_LSTM", False)
    dev = torch.device("cuda:0")
    D, lens = 128, [40, 3, 77, 18]
    x = torch.from_numpy(np.concatenate([R.features(T, 1, D, 70 + i)[:, 0, :] for i, T in enumerate(lens)])).to(dev)
    target = torch.rand(sum(lens), generator=torch.Generator().manual_seed(5)).to(dev)

    def make():
        torch.manual_seed(11)
        m = DSN(input_size=D, hidden_size=48).to(dev)
        return m, FlatAdam(m.parameters(), lr=1e-3, weight_decay=1e-5)

    def step(m, opt):
        opt.zero_grad()
        loss = torch.mean((m.score_packed(x, lens) - target) ** 2) * 50.0      # large enough for the 0.05 clip to bite
        loss.backward()
        opt.step(max_norm=0.05)
        return loss.detach()

    ma, oa = make()
    eager = [float(step(ma, oa)) for _ in range(4)]
    mb, ob = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        first = step(mb, ob)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step(mb, ob)
    got = [float(first)]
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        got.append(float(out))
    assert got == eager, (got, eager)
    assert int(ob._state[0]) == 4                                             # the device-side step counter followed the replays
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), k


def test_mfma_rate_probe_is_plausible_and_refuses_bad_arguments():
    """csrc/mfma_probe.hip (bench.py's `mfma_sustained`): the three MFMA kinds land between a tenth of and slightly above the guide's dense peaks (2 500
    TFLOP/s bf16, 157.3 f32), and a longer run gives the same rate (the figure is a rate, not a launch-overhead artefact)."""
    from summarizer_amd import _lib, kernels
    lib = _lib.load()
    for rnd in (False, True):
        for kind, peak in (("bf16", 2500.0), ("bf16_16", 2500.0), ("f32", 157.3)):
            t1, s1 = kernels.mfma_sustained_rate(kind, 2000, rnd)
            t2, s2 = kernels.mfma_sustained_rate(kind, 8000, rnd)
            assert 0.1 * peak < t1 < 1.05 * peak and 0.1 * peak < t2 < 1.05 * peak, (kind, rnd, t1, t2)
            assert abs(t1 - t2) < 0.25 * t2 and s2 > 2.5 * s1, (kind, rnd, t1, t2, s1, s2)
    t = C.c_double(0)
    assert lib.sumk_probe_mfma_rate(7, 10, C.byref(t), None, None) == -1
    assert lib.sumk_probe_mfma_rate(0, 0, C.byref(t), None, None) == -1
