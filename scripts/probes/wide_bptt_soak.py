"""Soak test of the persistent recurrences' hand-off protocol: the same sLSTM (H = 1024, 2 layers) training step is repeated N
times from identical inputs; every repetition's scores and gradients must equal the first one's BIT FOR BIT (the kernels are
deterministic, so any difference is a stale or torn read in a hand-off), and the health word must stay clear.  A second pass
runs a bandwidth hog on another stream between repetitions to warm the caches with foreign lines ("uneven load")."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import numpy as np, torch, recipes as R
from summarizer_amd import kernels
from summarizer_amd.models.sumgan import sLSTM
from summarizer_amd.models.dsn import DSN

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)]
x = torch.from_numpy(np.concatenate([R.features(T, 1, 1024, i)[:, 0, :] for i, T in enumerate(lens)])).to(dev)
cw = torch.randn(sum(lens), device=dev)
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
for name, make in (("sLSTM H=1024 x2", lambda: sLSTM(1024, 1024, 2)), ("DSN H=256", lambda: DSN(1024, 256, 1))):
    torch.manual_seed(1)
    m = make().to(dev)
    ref = None
    bad = 0
    for it in range(2 * N):
        if it >= N:
            junk.add_(1)                                  # foreign traffic: evicts / refills L2 and the Infinity Cache
        for p in m.parameters():
            p.grad = None
        s = m.score_packed(x, lens)
        (s * cw).sum().backward()
        cur = [s.detach().clone()] + [p.grad.detach().clone() for p in m.parameters()]
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
            bad += 1
    torch.cuda.synchronize()
    kernels.health_check()
    print(f"{name}: {2 * N} repetitions, {bad} differed from the first; health word clear")
    assert bad == 0
