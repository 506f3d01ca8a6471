# Probe: which torch-level operators launch kernels inside one bf16 VASNet training step (bench.py's run_step), by torch.profiler.
import sys, torch
sys.path.insert(0, ".")
import bench, numpy as np
from summarizer_amd.models.vasnet import VASNet
from summarizer_amd.training import FlatAdam
from summarizer_amd import kernels as _k
from summarizer_amd.autograd import SegmentMseMeanFunction
dev = torch.device("cuda:0")
lens = bench.tvsum_lens(50); frames = int(sum(lens))
x = (torch.randn(frames, 1024, device=dev).abs() * 0.5)
torch.manual_seed(0)
model = VASNet(input_size=1024).to(dev); model.precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"; model.train()
opt = FlatAdam(model.parameters(), lr=5e-5, weight_decay=1e-5, comm_dtype=torch.bfloat16)
target = torch.rand(frames, device=dev)
sb = _k.SeqBatch.get(lens, dev)
def step():
    opt.zero_grad()
    loss = SegmentMseMeanFunction.apply(model.score_packed(x, lens), target, sb, 1.0 / len(lens))
    loss.backward()
    opt.step(grad_scale=opt.all_reduce_grads(), zero_grad=True)
for _ in range(5): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
from collections import Counter
c = Counter(e.name for e in evs)
for k, v in c.most_common(25): print(f"{k:40s} {v/3:.1f} per step")
