"""Times csrc/gemm_b16.hip (bf16 operands in HBM) on the row-wise GEMM shapes of the S-TVSum training step, next to the plane kernel
(fp32 operands converted per k-tile, precision code 3) on the same shapes.  Prints us and TFLOP/s per shape."""
import ctypes as C
import torch
from summarizer_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
R, D = 12003, 1024
shapes = [("QKV  NT", 0, R, 3 * D, D), ("oproj NT", 0, R, D, D), ("dY1  NN", 1, R, D, D), ("dW   TN", 2, D, D, R), ("dWqkv TN", 2, 3 * D, D, R)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, layout, M, N, K in shapes:
    a = torch.randn((K, M) if layout == 2 else (M, K), device=dev)
    b = torch.randn((N, K) if layout == 0 else (K, N), device=dev)
    a16, b16 = a.to(torch.bfloat16), b.to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    ws = torch.zeros(8192 + 32 * M * N * 4, dtype=torch.uint8, device=dev)
    fl = 2.0 * M * N * K
    t_new = timed(lambda: _lib.check(lib.sumk_gemm_bf16src(layout, a16.data_ptr(), b16.data_ptr(), c.data_ptr(), M, N, K, None, 0, st), "b16"))
    line = f"{name}: bf16-source {t_new:7.1f} us {fl / t_new / 1e6:7.1f} TF/s"
    if layout == 2:
        t_sk = timed(lambda: _lib.check(lib.sumk_gemm_bf16src(2, a16.data_ptr(), b16.data_ptr(), c.data_ptr(), M, N, K, ws.data_ptr(), ws.numel(), st), "b16sk"))
        line += f" | split-K {t_sk:7.1f} us {fl / t_sk / 1e6:7.1f} TF/s"
    t_old = timed(lambda: _lib.check(lib.sumk_gemm_prec(layout, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 3, st), "plane"))
    line += f" | plane kernel {t_old:7.1f} us {fl / t_old / 1e6:7.1f} TF/s"
    print(line, flush=True)
