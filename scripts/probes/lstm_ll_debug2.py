import sys, numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
from summarizer_amd import kernels
from summarizer_amd.models.dsn import DSN
torch.manual_seed(0)
In, H, lens = 64, 16, [37]
m = DSN(In, H, 1).eval().to("cuda:0")
x = torch.randn(sum(lens), In, device="cuda:0")
sb = kernels.SeqBatch.get(lens, torch.device("cuda:0"))
P = dict(m.named_parameters())
h, _ = kernels.bilstm_layer_forward(x, sb, P, "rnn.", 0, H)
h = h.detach().cpu().numpy()
# torch reference, and the same with h_{t-1} forced to zero / to the value TWO steps back
lstm = torch.nn.LSTM(In, H, 1, bidirectional=True)
sd = {k.replace("rnn.", ""): v.detach().cpu() for k, v in P.items() if k.startswith("rnn.")}
lstm.load_state_dict(sd)
ref = lstm(x.cpu().unsqueeze(1))[0][:, 0, :].detach().numpy()
print("vs torch: max err per row (fwd half), rows 0..5:", np.abs(h - ref)[:6, :H].max(axis=1))
W_ih, W_hh, b = sd["weight_ih_l0"].numpy(), sd["weight_hh_l0"].numpy(), (sd["bias_ih_l0"] + sd["bias_hh_l0"]).numpy()
def sig(v): return 1 / (1 + np.exp(-v))
def run(mode):
    hs, c, hp, hpp = [], np.zeros(H), np.zeros(H), np.zeros(H)
    for t in range(lens[0]):
        hin = {"true": hp, "zero": np.zeros(H), "two_back": hpp}[mode]
        g = W_ih @ x[t].cpu().numpy() + W_hh @ hin + b
        i, f, gg, o = sig(g[:H]), sig(g[H:2*H]), np.tanh(g[2*H:3*H]), sig(g[3*H:])
        c = f * c + i * gg; hn = o * np.tanh(c); hs.append(hn); hpp = hp; hp = hn
    return np.array(hs)
for mode in ("true", "zero", "two_back"):
    print(mode, "row1 err", np.abs(run(mode)[1] - h[1, :H]).max(), "row2 err", np.abs(run(mode)[2] - h[2, :H]).max())
import itertools
h0 = run("true")[0]
def step1(hin):
    g = W_ih @ x[1].cpu().numpy() + W_hh @ hin + b
    c0 = None
    # recompute c after step 0
    g0 = W_ih @ x[0].cpu().numpy() + b
    c = sig(g0[:H]) * np.tanh(g0[2*H:3*H])
    i, f, gg, o = sig(g[:H]), sig(g[H:2*H]), np.tanh(g[2*H:3*H]), sig(g[3*H:])
    c = f * c + i * gg
    return o * np.tanh(c)
best = []
for mask in range(16):
    hin = h0.copy()
    for q in range(4):
        if not (mask >> q) & 1: hin[4*q:4*q+4] = 0
    best.append((np.abs(step1(hin) - h[1, :H]).max(), "chunks present %s" % bin(mask)))
for perm in itertools.permutations(range(4)):
    hin = h0.reshape(4, 4)[:, list(perm)].reshape(-1)
    best.append((np.abs(step1(hin) - h[1, :H]).max(), "within-chunk perm %s" % (perm,)))
for perm in itertools.permutations(range(4)):
    hin = h0.reshape(4, 4)[list(perm), :].reshape(-1)
    best.append((np.abs(step1(hin) - h[1, :H]).max(), "chunk perm %s" % (perm,)))
best.sort()
print(best[:5])
