"""numpy restatement of the bidirectional-LSTM scorers.  TEST INFRASTRUCTURE ONLY.

Reference: DSN.forward  summarizer/models/dsn.py:38-47  (nn.LSTM(1024,256,bidirectional) -> Linear(512,1) -> Sigmoid)
           sLSTM.forward summarizer/models/sumgan.py:36-46 (nn.LSTM(1024,1024,2 layers,bidirectional) -> Linear -> Sigmoid)
LSTM cell math is torch.nn.LSTM's documented recurrence (gate order i,f,g,o; two bias vectors; h0=c0=0),
confirmed against the reference modules in tests/golden/make_golden.py.
Weights: dict keyed like the state_dict, prefix 'rnn.' (DSN) or 'lstm.' (sLSTM):
  <p>weight_ih_l{k}[_reverse] (4H,in), <p>weight_hh_l{k}[_reverse] (4H,H), <p>bias_ih_l{k}[_reverse], <p>bias_hh_l{k}[_reverse]
"""
import numpy as np


def _sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse, f=np.float32):
    """x: (T,B,In) -> h: (T,B,H).  One direction of one layer."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    G = x @ w_ih.T + (b_ih + b_hh)              # hoisted input projection (T,B,4H)
    h = np.zeros((B, H), dtype=f)
    c = np.zeros((B, H), dtype=f)
    out = np.empty((T, B, H), dtype=f)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = G[t] + h @ w_hh.T
        i = _sigmoid(g[:, 0 * H:1 * H])
        fg = _sigmoid(g[:, 1 * H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H])
        o = _sigmoid(g[:, 3 * H:4 * H])
        c = fg * c + i * gg
        h = o * np.tanh(c)
        out[t] = h
    return out


def bilstm_forward(x, p, prefix, num_layers, dtype=np.float32):
    """x: (T,B,D) -> (T,B,2H): stacked bidirectional LSTM, layer k+1 fed [fwd||bwd] of layer k."""
    f = dtype
    inp = np.asarray(x).astype(f)
    for k in range(num_layers):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            g = lambda n: np.asarray(p[f"{prefix}{n}_l{k}{suffix}"]).astype(f)
            outs.append(lstm_direction(inp, g("weight_ih"), g("weight_hh"), g("bias_ih"), g("bias_hh"), rev, f))
        inp = np.concatenate(outs, axis=2)
    return inp


def dsn_forward(x, p, num_layers=1, dtype=np.float32):
    """dsn.py:38-47.  p keys: rnn.*, out.0.weight (1,2H), out.0.bias (1,)"""
    h = bilstm_forward(x, p, "rnn.", num_layers, dtype)
    s = h @ np.asarray(p["out.0.weight"]).astype(dtype).T + np.asarray(p["out.0.bias"]).astype(dtype)
    return _sigmoid(s)


def slstm_forward(x, p, num_layers=2, dtype=np.float32):
    """sumgan.py:36-46.  p keys: lstm.*, out.weight (1,2H), out.bias (1,)"""
    h = bilstm_forward(x, p, "lstm.", num_layers, dtype)
    s = h @ np.asarray(p["out.weight"]).astype(dtype).T + np.asarray(p["out.bias"]).astype(dtype)
    return _sigmoid(s)
