"""Seeded input/weight recipes shared by the golden generator (make_golden.py, run once in the build
container against the real reference) and by the tests (which regenerate the same inputs anywhere).
numpy's PCG64 `default_rng` stream is platform independent, so full-size (D=1024) cases need only their
expected OUTPUTS committed, not 21 MB of weights."""
import hashlib
import numpy as np

VASNET_KEYS = ["K.weight", "Q.weight", "V.weight", "attention_head_projection.weight",
               "k1.weight", "k1.bias", "k2.weight", "k2.bias", "layer_norm.weight", "layer_norm.bias"]


def vasnet_weights(D, seed, max_length=None):
    """Xavier-like uniform matrices, non-trivial biases and LN affine (so the shared-LN quirk is exercised)."""
    rng = np.random.default_rng(seed)
    a = np.sqrt(2.0) * np.sqrt(6.0 / (2 * D))
    p = {}
    for k in ["K.weight", "Q.weight", "V.weight", "attention_head_projection.weight", "k1.weight"]:
        p[k] = rng.uniform(-a, a, (D, D)).astype(np.float32)
    p["k1.bias"] = rng.uniform(-0.2, 0.2, (D,)).astype(np.float32)
    p["k2.weight"] = rng.uniform(-a * 4, a * 4, (1, D)).astype(np.float32)
    p["k2.bias"] = rng.uniform(-0.2, 0.2, (1,)).astype(np.float32)
    p["layer_norm.weight"] = rng.uniform(0.5, 1.5, (D,)).astype(np.float32)
    p["layer_norm.bias"] = rng.uniform(-0.1, 0.1, (D,)).astype(np.float32)
    if max_length:
        p["pos_embed.weight"] = rng.normal(0, 1, (max_length, D)).astype(np.float32)
    return p


def lstm_weights(prefix, D, H, num_layers, seed, head_prefix):
    rng = np.random.default_rng(seed)
    k = 1.0 / np.sqrt(H)
    p = {}
    for l in range(num_layers):
        In = D if l == 0 else 2 * H
        for suf in ("", "_reverse"):
            p[f"{prefix}weight_ih_l{l}{suf}"] = rng.uniform(-k, k, (4 * H, In)).astype(np.float32)
            p[f"{prefix}weight_hh_l{l}{suf}"] = rng.uniform(-k, k, (4 * H, H)).astype(np.float32)
            p[f"{prefix}bias_ih_l{l}{suf}"] = rng.uniform(-k, k, (4 * H,)).astype(np.float32)
            p[f"{prefix}bias_hh_l{l}{suf}"] = rng.uniform(-k, k, (4 * H,)).astype(np.float32)
    kh = 1.0 / np.sqrt(2 * H)
    p[f"{head_prefix}weight"] = rng.uniform(-kh * 3, kh * 3, (1, 2 * H)).astype(np.float32)
    p[f"{head_prefix}bias"] = rng.uniform(-kh, kh, (1,)).astype(np.float32)
    return p


def transformer_weights(D, n_layers, seed, max_length=None):
    """Weights of the reference Transformer scorer (state_dict keys of summarizer/models/transformer.py), seeded."""
    rng = np.random.default_rng(seed)
    a = np.sqrt(6.0 / (2 * D))
    u = lambda *shape, s=1.0: rng.uniform(-a * s, a * s, shape).astype(np.float32)
    p = {}
    for l in range(n_layers):
        pre = f"transformer_encoder.layers.{l}."
        p[pre + "self_attn.in_proj_weight"] = u(3 * D, D, s=1.5)
        p[pre + "self_attn.in_proj_bias"] = rng.uniform(-0.1, 0.1, (3 * D,)).astype(np.float32)
        p[pre + "self_attn.out_proj.weight"] = u(D, D)
        p[pre + "self_attn.out_proj.bias"] = rng.uniform(-0.1, 0.1, (D,)).astype(np.float32)
        p[pre + "linear1.weight"] = u(D, D, s=1.4); p[pre + "linear1.bias"] = rng.uniform(-0.1, 0.1, (D,)).astype(np.float32)
        p[pre + "linear2.weight"] = u(D, D, s=1.4); p[pre + "linear2.bias"] = rng.uniform(-0.1, 0.1, (D,)).astype(np.float32)
        for n in ("norm1", "norm2"):
            p[pre + n + ".weight"] = rng.uniform(0.7, 1.3, (D,)).astype(np.float32)
            p[pre + n + ".bias"] = rng.uniform(-0.1, 0.1, (D,)).astype(np.float32)
    p["layer_norm.weight"] = rng.uniform(0.7, 1.3, (D,)).astype(np.float32)
    p["layer_norm.bias"] = rng.uniform(-0.1, 0.1, (D,)).astype(np.float32)
    p["transformer_encoder.norm.weight"] = p["layer_norm.weight"]          # the same module registered twice
    p["transformer_encoder.norm.bias"] = p["layer_norm.bias"]
    p["k1.weight"] = u(D, D, s=1.4); p["k1.bias"] = rng.uniform(-0.2, 0.2, (D,)).astype(np.float32)
    p["k2.weight"] = u(1, D, s=4.0); p["k2.bias"] = rng.uniform(-0.2, 0.2, (1,)).astype(np.float32)
    if max_length:
        p["pos_embed.weight"] = rng.normal(0, 1, (max_length, D)).astype(np.float32)
    return p


def features(T, B, D, seed):
    """pool5-like non-negative features: 0.5*|N(0,1)|."""
    rng = np.random.default_rng(seed)
    return (0.5 * np.abs(rng.standard_normal((T, B, D)))).astype(np.float32)


def digest(arrs):
    h = hashlib.sha256()
    for k in sorted(arrs):
        h.update(k.encode()); h.update(np.ascontiguousarray(arrs[k]).tobytes())
    return h.hexdigest()


def synthetic_video(T, seed, n_users=15, D=None):
    """SumMe/TVSum-shaped evaluation metadata: picks every 15th frame, random change points."""
    rng = np.random.default_rng(seed)
    n_frames = int(15 * T - rng.integers(0, 15))
    picks = (15 * np.arange(T)).astype(np.int32)
    n_seg = max(1, T // 15)
    cuts = np.sort(rng.choice(np.arange(15, n_frames - 15), size=n_seg - 1, replace=False)) if n_seg > 1 else np.array([], int)
    starts = np.concatenate([[0], cuts]).astype(np.int64)
    ends = np.concatenate([cuts - 1, [n_frames - 1]]).astype(np.int64)
    cps = np.stack([starts, ends], axis=1).astype(np.int32)
    nfps = (ends - starts + 1).astype(np.int32)
    user_summary = (rng.random((n_users, n_frames)) < 0.15).astype(np.float32)
    gt = rng.random((T + 1) // 2).repeat(2)[:T].astype(np.float32)
    user_scores = np.clip(gt[None, :] + 0.3 * rng.standard_normal((n_users, T)), 0, 1).astype(np.float32)
    user_scores_frames = np.repeat(user_scores, 15, axis=1)[:, :n_frames]
    if user_scores_frames.shape[1] < n_frames:
        user_scores_frames = np.pad(user_scores_frames, ((0, 0), (0, n_frames - user_scores_frames.shape[1])))
    d = dict(n_frames=n_frames, picks=picks, change_points=cps, n_frame_per_seg=nfps,
             user_summary=user_summary, gtscore=gt, user_scores=user_scores_frames.astype(np.float32))
    if D:
        d["features"] = features(T, 1, D, seed + 7)[:, 0, :]
    return d


def dropout_keep(seed, site, idx, p):
    """numpy twin of sumk::dropout_keep (csrc/sumk_internal.h): keep-mask of training-mode dropout as a pure
    function of (seed, site, element index).  idx: uint64 array.  Returns a bool array (True = kept)."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) ^ (np.uint64(site + 1) * np.uint64(0x9E3779B97F4A7C15))) + idx * np.uint64(0xD1342543DE82EF95)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    thr = min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF)
    return (z >> np.uint64(32)).astype(np.uint64) >= np.uint64(thr)


def vasnet_drop_masks(seed, p, lens, D):
    """Scaled keep-masks (0 or 1/(1-p)) of the three dropout sites for a packed batch, per video:
    returns list of (m_alpha (T,T), m_y (T,D), m_z (T,D)) float32, indexed exactly like the HIP kernels."""
    sc = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    out, row0 = [], 0
    for T in lens:
        rows = (np.arange(T, dtype=np.uint64) + np.uint64(row0))
        ia = (rows[:, None] << np.uint64(20)) | np.arange(T, dtype=np.uint64)[None, :]
        iy = rows[:, None] * np.uint64(D) + np.arange(D, dtype=np.uint64)[None, :]
        out.append(tuple((dropout_keep(seed, site, ix, p).astype(np.float32) * sc) for site, ix in ((0, ia), (1, iy), (2, iy))))
        row0 += T
    return out


class DetRandom:
    """Counter-based stand-in for torch.randn_like / torch.rand, used to run the reference's SumGAN trainer and the HIP one
    on the SAME draws (their generators -- torch CPU vs device -- cannot be matched): draw n comes from
    numpy.random.default_rng([seed, n]) whatever the device."""

    def __init__(self, seed):
        self.seed, self.n = int(seed), 0

    def _next(self, shape, kind):
        import torch
        self.n += 1
        g = np.random.default_rng([self.seed, self.n])
        a = g.standard_normal(tuple(shape)) if kind == "normal" else g.random(tuple(shape))
        return torch.from_numpy(a.astype(np.float32))

    def randn_like(self, t, **kw):
        return self._next(t.shape, "normal").to(t.device)

    def rand(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list)):
            size = tuple(size[0])
        out = self._next(size, "uniform")
        return out.to(kw["device"]) if kw.get("device") is not None else out

    def patch(self):
        """Context manager: torch.randn_like / torch.rand -> this generator."""
        import contextlib
        import torch

        @contextlib.contextmanager
        def cm():
            old = torch.randn_like, torch.rand
            torch.randn_like, torch.rand = self.randn_like, self.rand
            try:
                yield self
            finally:
                torch.randn_like, torch.rand = old
        return cm()


def sample_idx(name, numel, n=256):
    """Flat indices at which a large tensor is digested (train_full.npz): seeded by the parameter name, sorted, unique."""
    seed = int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")
    return np.sort(np.random.default_rng(seed).choice(numel, size=min(n, numel), replace=False))
