import sys, numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import recipes as R
from summarizer_amd.models.dsn import DSN
D, H, L = 128, int(sys.argv[2]), 1
lens = [int(v) for v in sys.argv[3].split(",")]
w = R.lstm_weights("rnn.", D, H, L, 21, "out.0.")
m = DSN(D, H, L); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to("cuda:0")
xs = [R.features(T, 1, D, 80 + i) - 0.2 for i, T in enumerate(lens)]
xp = torch.from_numpy(np.concatenate([x[:, 0, :] for x in xs])).to("cuda:0").requires_grad_(True)
s = m.score_packed(xp, lens)
cw = torch.from_numpy(np.random.default_rng(4).standard_normal(sum(lens)).astype(np.float32)).to("cuda:0")
(s * cw).sum().backward()
np.savez(sys.argv[1], gx=xp.grad.cpu().numpy(), **{k: p.grad.cpu().numpy() for k, p in m.named_parameters()})
from summarizer_amd import kernels as _k
import time
try:
    _k.health_check(); print("health ok")
except Exception as e:
    print("HEALTH:", str(e)[:200])
