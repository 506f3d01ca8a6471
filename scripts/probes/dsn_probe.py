"""Probe of DSN scoring on the S-TVSum batch per precision: whole call and the recurrence launch alone (sumk_prof_*).
SUMK_LSTM_PROJ=1 runs the projection inside the recurrence (lstm_persist_proj_kernel); default: the plane GEMM in front."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from summarizer_amd import _lib, kernels
from summarizer_amd.models.dsn import DSN

dev = torch.device("cuda:0")
lib = _lib.load()
rng = np.random.default_rng(0)
lens = [int(v) for v in rng.integers(150, 321, 50)]
lens[0] = 320
frames = sum(lens)
torch.manual_seed(1234)
x = torch.randn(frames, 1024, device=dev) * 0.5
m = DSN(input_size=1024).to(dev).eval()
out = {}
for prec in ["fp32", "bf16x6", "bf16x3"]:
    m.precision = prec
    with torch.no_grad():
        for _ in range(5):
            s = m.score_packed(x, lens)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            s = m.score_packed(x, lens)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        lib.sumk_prof_enable(1 << 2)
        for _ in range(n):
            s = m.score_packed(x, lens)
        ms, cnt = C.c_double(0), C.c_int64(0)
        lib.sumk_prof_read(2, C.byref(ms), C.byref(cnt), 1)
        lib.sumk_prof_enable(0)
    kernels.health_check()
    out[prec] = s.cpu().numpy()
    print(f"{prec:7s} PROJ={os.environ.get('SUMK_LSTM_PROJ', '0')}: whole call {dt * 1e3:7.3f} ms; recurrence launch {ms.value / max(cnt.value, 1) * 1e3:8.1f} us "
          f"= {ms.value / max(cnt.value, 1) * 1e3 / max(lens):6.2f} us per step", flush=True)
print("max |bf16x6 - fp32| =", float(np.abs(out["bf16x6"] - out["fp32"]).max()), " max |bf16x3 - fp32| =", float(np.abs(out["bf16x3"] - out["fp32"]).max()))
