"""Debug probe: wide persistent recurrence vs the CPU oracle, one layer, per step / direction / unit block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import numpy as np, torch
import recipes as R
from summarizer_amd import kernels
from oracle import lstm_np
D, H = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lens = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5]
dev = torch.device("cuda:0")
w = R.lstm_weights("rnn.", D, H, 1, 5, "out.0.")
p = {k: torch.from_numpy(v).to(dev) for k, v in w.items()}
xs = [R.features(T, 1, D, 30 + i) - 0.2 for i, T in enumerate(lens)]
x = torch.from_numpy(np.concatenate([v[:, 0, :] for v in xs])).to(dev)
sb = kernels.SeqBatch.get(lens, dev)
h, _ = kernels.bilstm_layer_forward(x, sb, p, "rnn.", 0, H)
h = h.cpu().numpy()
off = np.concatenate([[0], np.cumsum(lens)])
for i, xv in enumerate(xs):
    ref = lstm_np.bilstm_forward(xv, w, 'rnn.', 1)[:, 0, :]
    got = h[off[i]:off[i + 1]]
    for t in list(range(min(lens[i], 3))) + list(range(max(3, lens[i] - 3), lens[i])):
        for d in range(2):
            e = np.abs(got[t, d * H:(d + 1) * H] - ref[t, d * H:(d + 1) * H])
            print(f"video {i} t={t} dir={d}: max err {e.max():.3e} at unit {e.argmax()}  first bad units {np.nonzero(e > 1e-4)[0][:8]}")
# hypothesis check at t=1, dir 0, video 0: which part of the recurrent term is present?
xv = xs[0][:, 0, :]
g = lambda n: w[f"rnn.{n}_l0"]
G = xv @ g("weight_ih").T + g("bias_ih") + g("bias_hh")
ref = lstm_np.bilstm_forward(xs[0], w, 'rnn.', 1)[:, 0, :]
h0 = ref[0, :H]
def step(pre, c_prev):
    sg = lambda v: 1 / (1 + np.exp(-v))
    i, f, gg, o = sg(pre[:H]), sg(pre[H:2*H]), np.tanh(pre[2*H:3*H]), sg(pre[3*H:])
    c = f * c_prev + i * gg
    return o * np.tanh(c), c
pre0 = G[0]
_, c0 = step(pre0, np.zeros(H, np.float32))
got = h[1, :H]
Whh = g("weight_hh")
for name, rec in (("full", h0 @ Whh.T), ("none", 0 * G[1]), ("half-lo k", h0[:H//2] @ Whh[:, :H//2].T), ("half-hi k", h0[H//2:] @ Whh[:, H//2:].T)):
    hh, _ = step(G[1] + rec, c0)
    print(name, "max |got - hyp| =", float(np.abs(got - hh).max()))
for wv in range(8):
    ks = slice(wv * H // 8, (wv + 1) * H // 8)
    hh, _ = step(G[1] + h0[ks] @ Whh[:, ks].T, c0)
    print("only wave", wv, "k-range:", float(np.abs(got - hh).max()))
dbg = int(os.environ.get("SUMK_WIDE_DBG", "0"))
if 3 <= dbg < 10:
    q = dbg - 3
    rec = h0 @ Whh[q * H:(q + 1) * H].T
    got_rec = h[1, :H]
    print("recurrent term gate", q, ": max err", float(np.abs(got_rec - rec).max()))
    print(" ref ", rec[:12]); print(" got ", got_rec[:12])
    print(" ratio", (got_rec / rec)[:12])
    for name, alt in (("W other dir", h0 @ w["rnn.weight_hh_l0_reverse"][q*H:(q+1)*H].T), ("h of reverse dir row0", ref[0, H:] @ Whh[q*H:(q+1)*H].T)):
        print(name, float(np.abs(got_rec - alt).max()))
if 3 <= dbg < 10:
    allrec = h0 @ Whh.T          # (4H,)
    for j in list(range(12)) + [100, 530, 1023]:
        m = np.abs(allrec - got_rec[j]); jj = int(m.argmin())
        print(f"got[{j}] = {got_rec[j]:+.6f} nearest full-rec index {jj} (gate {jj // H}, unit {jj % H}) err {m.min():.2e}")
    # k-subset hypotheses
    kk = np.arange(H)
    for name, mask in (("k%8<4", kk % 8 < 4), ("k%8>=4", kk % 8 >= 4), ("k<512", kk < 512)):
        alt = (h0 * mask) @ Whh[q * H:(q + 1) * H].T
        print(name, float(np.abs(got_rec - alt).max()), float(np.abs(got_rec - 2 * alt).max()))
if dbg == 13:
    alt = Whh[:H].sum(1); print("A=1: max err vs row sums of W", float(np.abs(h[1, :H] - alt).max())); print(alt[:8]); print(h[1, :8])
if dbg == 23:
    print("B=1: expect", float(h0.sum()), "got", h[1, :16])
if dbg == 33:
    A = h[3, :H]
    e = np.abs(A - h0)
    print("A dump vs h0: max err", float(e.max()), "n bad", int((e > 1e-6).sum()), "bad idx", np.nonzero(e > 1e-6)[0][:40])
    print("A[:16]", A[:16]); print("h0[:16]", h0[:16])
