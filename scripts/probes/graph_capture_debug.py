"""Probe: capture a packed scoring call in a HIP graph (torch.cuda.CUDAGraph) and compare replays with eager results."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import recipes as R
from summarizer_amd.models.vasnet import VASNet
from summarizer_amd.models.dsn import DSN
dev = torch.device("cuda:0")
D, lens = 256, [70, 1, 33, 129, 64]
for name, make in (("vasnet", lambda: VASNet(input_size=D)), ("dsn", lambda: DSN(input_size=D, hidden_size=32))):
    torch.manual_seed(3)
    m = make().to(dev).eval()
    xs = [torch.from_numpy(np.concatenate([R.features(T, 1, D, 10 * k + i)[:, 0, :] for i, T in enumerate(lens)])).to(dev) for k in range(2)]
    with torch.no_grad():
        eager = [m.score_packed(x, lens).clone() for x in xs]
        static_x = xs[0].clone()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.score_packed(static_x, lens)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m.score_packed(static_x, lens)
        for k in (0, 1, 0):
            static_x.copy_(xs[k]); g.replay(); torch.cuda.synchronize()
            print(name, "replay", k, "max|d| vs eager", float((out - eager[k]).abs().max()), "nan", bool(torch.isnan(out).any()))
