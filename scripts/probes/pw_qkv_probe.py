"""Probe: the QKV launch of the bf16x6 / bf16x3 scoring step (planes out) on the S-TVSum batch, HIP events on the launch stream (sumk_prof_*),
alternating nothing -- run it once per library (SUMK_LIB_PATH) on the same box; prints us per launch and the step."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import bench, recipes as R
from summarizer_amd import _lib
from summarizer_amd.models.vasnet import VASNet
dev = torch.device("cuda:0"); lib = _lib.load()
lens = bench.tvsum_lens(50)
x = torch.from_numpy(np.concatenate([R.features(T, 1, 1024, i)[:, 0, :] for i, T in enumerate(lens)])).to(dev)
torch.manual_seed(1234)
m = VASNet(input_size=1024).to(dev).eval()
for prec in ("bf16x6", "bf16x3"):
    m.precision = prec
    with torch.no_grad():
        for _ in range(30): m.score_packed(x, lens)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): m.score_packed(x, lens)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        lib.sumk_prof_read(0, None, None, 1); lib.sumk_prof_enable(1)
        for _ in range(30): m.score_packed(x, lens)
        torch.cuda.synchronize(); lib.sumk_prof_enable(0)
        ms, n = C.c_double(0), C.c_int64(0); lib.sumk_prof_read(0, C.byref(ms), C.byref(n), 1)
    print(f"{prec} lib={os.path.basename(os.environ.get('SUMK_LIB_PATH', 'libsumk.so'))}: step {dt * 1e3:.4f} ms, QKV launch {ms.value / n.value * 1e3:.1f} us", flush=True)
