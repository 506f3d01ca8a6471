"""numpy restatement of DSNTrainer.compute_reward (reference: summarizer/models/dsn.py:185-236).  TEST INFRASTRUCTURE ONLY."""
import numpy as np


def compute_reward(seq, actions, far_sim=False, temp_dist_thre=20, dtype=np.float32):
    """seq: (T,1,D) or (T,D); actions: (T,1,1) or (T,) binary.  Returns a python float-like scalar (dtype)."""
    f = dtype
    x = np.asarray(seq).reshape(np.asarray(seq).shape[0], -1).astype(f)       # dsn.py:207
    a = np.asarray(actions).reshape(-1)
    picks = np.nonzero(a)[0]                                                   # dsn.py:195
    n = len(picks)
    if n == 0:                                                                 # dsn.py:199-203
        return f(0.0)
    T = x.shape[0]
    if n == 1:                                                                 # dsn.py:211-214
        r_div = f(0.0)
    else:
        normed = x / np.sqrt((x * x).sum(axis=1, keepdims=True))               # dsn.py:217
        dissim = f(1.0) - normed @ normed.T                                    # dsn.py:218
        sub = dissim[picks][:, picks].copy()                                   # dsn.py:219
        if not far_sim:
            td = np.abs(picks[None, :] - picks[:, None])                       # dsn.py:222-223
            sub[td > temp_dist_thre] = f(1.0)                                  # dsn.py:224
        r_div = sub.sum(dtype=f) / f(n * (n - 1.0))                            # dsn.py:225
    sq = (x * x).sum(axis=1, keepdims=True)                                    # dsn.py:228
    dist = sq + sq.T - f(2.0) * (x @ x.T)                                      # dsn.py:229-230
    dist = dist[:, picks].min(axis=1)                                          # dsn.py:231-232
    r_rep = np.exp(-dist.mean(dtype=f))                                        # dsn.py:233
    return f((r_div + r_rep) * f(0.5))                                         # dsn.py:236
