#!/usr/bin/env python3
"""Headline batch (S-TVSum, 50 videos, D = 1024, exact fp32): step time and the Q.K^T / alpha.V / QKV launch durations under
environment knobs (SUMK_LEAN, SUMK_LEAN128, SUMK_LEAN_GRID, SUMK_GROUP_REMAP, SUMK_ATTN_CFG ...).  The knobs are read once per process,
so the sweep re-runs itself as a child per configuration:
    python scripts/probes/attn_balance_sweep.py "SUMK_LEAN=0" "SUMK_LEAN=1,SUMK_GROUP_REMAP=2"
(round 3 used it for the experiments DESIGN.md section 3 lists: residency caps, dynamic tile queue, one-barrier loop, wave priorities,
staggered slots, heaviest-first tile order -- all since removed -- and for the lean kernels that stayed)."""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def child():
    import numpy as np, torch
    import recipes as R
    from summarizer_amd import _lib
    from summarizer_amd.models.vasnet import VASNet
    lib = _lib.load()
    dev = torch.device("cuda:0")
    lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)]
    torch.manual_seed(1234)
    m = VASNet(input_size=1024).to(dev).eval()
    m.precision = os.environ.get("PROBE_PRECISION", "fp32")
    x = torch.from_numpy(np.concatenate([R.features(T, 1, 1024, i)[:, 0, :] for i, T in enumerate(lens)])).to(dev)
    with torch.no_grad():
        for _ in range(20):
            s = m.score_packed(x, lens)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(100):
                s = m.score_packed(x, lens)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100)
        out = dict(step_ms=round(best * 1e3, 4), checksum=float(s.double().sum()))
        for name, tag in (("qkt", _lib.PROF_GEMM_QKT), ("pv", _lib.PROF_GEMM_PV), ("qkv", _lib.PROF_GEMM_QKV)):
            lib.sumk_prof_read(tag, None, None, 1); lib.sumk_prof_enable(1 << tag)
            for _ in range(20):
                m.score_packed(x, lens)
            torch.cuda.synchronize()
            ms = C.c_double(0); n = C.c_int64(0)
            lib.sumk_prof_read(tag, C.byref(ms), C.byref(n), 1)
            out[name + "_us"] = round(ms.value / max(n.value, 1) * 1e3, 1)
        lib.sumk_prof_enable(0)
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD") == "1":
        child(); sys.exit(0)
    configs = [dict(SUMK_LEAN="0", SUMK_LEAN128="0"), dict(SUMK_LEAN="1", SUMK_LEAN128="0"), dict(SUMK_LEAN="1", SUMK_LEAN128="1")]
    if len(sys.argv) > 1:
        configs = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
    for cfg in configs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PROBE_CHILD="1", **cfg), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(cfg, line[-1] if line else ("FAILED " + r.stderr[-400:]), flush=True)
