#!/usr/bin/env python3
"""What the vendor library (hipBLASLt through torch.matmul, bf16 in / bf16 out) reaches on the GEMM shapes of the bf16 training step on this
box: the yardstick for csrc/gemm_b16.hip's wide kernel.  Not part of the product path."""
import torch, time
dev = torch.device("cuda:0")
R, D = 12003, 1024
def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = torch.randn(R, D, device=dev).bfloat16(); wqkv = torch.randn(3 * D, D, device=dev).bfloat16(); w = torch.randn(D, D, device=dev).bfloat16()
dq = torch.randn(R, 3 * D, device=dev).bfloat16(); dy = torch.randn(R, D, device=dev).bfloat16()
for name, fn, flop in [("QKV  (R,D)x(3D,D)^T", lambda: x @ wqkv.t(), 2 * R * D * 3 * D), ("proj (R,D)x(D,D)^T", lambda: x @ w.t(), 2 * R * D * D),
                       ("dX   (R,D)x(D,D)", lambda: dy @ w, 2 * R * D * D), ("dWqkv (3D,R)x(R,D)", lambda: dq.t() @ x, 2 * R * D * 3 * D),
                       ("dW   (D,R)x(R,D)", lambda: dy.t() @ x, 2 * R * D * D)]:
    us = t(fn)
    print(f"{name:24s} {us:7.1f} us  {flop / us / 1e9:6.3f} PFLOP/s  {flop / us / 1e9 / 2.5:5.2f} of the dense bf16 peak")
# the per-video products as the library's batched GEMM on videos padded to T = 320 (50 videos; the real batch has 12 003 of the 16 000 rows)
B, T = 50, 320
q = torch.randn(B, T, D, device=dev).bfloat16(); k = torch.randn(B, T, D, device=dev).bfloat16(); v = torch.randn(B, T, D, device=dev).bfloat16()
p = torch.randn(B, T, T, device=dev).bfloat16()
for name, fn, flop in [("Q.K^T  bmm (T,D)x(T,D)^T", lambda: torch.bmm(q, k.transpose(1, 2)), 2 * B * T * T * D), ("alpha.V bmm (T,T)x(T,D)", lambda: torch.bmm(p, v), 2 * B * T * T * D),
                       ("P^T.dC bmm (T,T)^Tx(T,D)", lambda: torch.bmm(p.transpose(1, 2), v), 2 * B * T * T * D)]:
    us = t(fn)
    print(f"{name:24s} {us:7.1f} us  {flop / us / 1e9:6.3f} PFLOP/s  {flop / us / 1e9 / 2.5:5.2f} of the dense bf16 peak")
