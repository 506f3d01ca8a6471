"""Probe: sLSTM (2-layer BiLSTM, H = 1024) MSE training step on the S-TVSum batch, ms per step (forward with saves + wide BPTT + GEMMs + Adam)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from summarizer_amd import kernels
from summarizer_amd.models.sumgan import sLSTM
from summarizer_amd.training import FlatAdam
from summarizer_amd.autograd import SegmentMseMeanFunction
dev = torch.device("cuda:0")
lens = bench.tvsum_lens(50); frames = sum(lens)
torch.manual_seed(1234)
x = torch.randn(frames, 1024, device=dev) * 0.5
m = sLSTM(input_size=1024).to(dev).train()
opt = FlatAdam(m.parameters(), lr=1e-5, weight_decay=1e-5)
target = torch.rand(frames, device=dev)
sb = kernels.SeqBatch.get(lens, dev)
def step():
    opt.zero_grad(zeroed_by_step=True)
    loss = SegmentMseMeanFunction.apply(m.score_packed(x, lens), target, sb, 1.0 / len(lens))
    loss.backward(gradient=kernels.one(dev))
    opt.step(grad_scale=1.0, zero_grad=True)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print(f"sLSTM training step R6={os.environ.get('SUMK_LSTM_BWD_R6', '1')} WIDE2={os.environ.get('SUMK_LSTM_WIDE2', '1')}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
kernels.health_check()
