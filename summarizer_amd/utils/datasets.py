"""Dataset access behind the h5py mapping protocol the reference uses
(`ds[key]["features"][...]`, `["n_frames"][()]`, `name in group`, `ds.keys()`; summarizer/models/__init__.py:15,47,70-79,
99-110,154-162; schema summarizer/datasets/README.md:5-42).

`open_dataset(path)` returns an h5py.File when h5py is importable and the path is an HDF5 file; otherwise a dict-backed
object with the same protocol (`DictDataset`), loadable from an .npz whose keys are "<video>/<field>".  The synthetic
SumMe/TVSum-shaped sets used by tests and bench.py are DictDatasets (no datasets can be downloaded here)."""
import numpy as np


class _Leaf:
    """Mimics an h5py Dataset: `leaf[...]` -> array copy, `leaf[()]` -> scalar/array."""
    def __init__(self, value):
        self._v = np.asarray(value)

    def __getitem__(self, idx):
        if idx is Ellipsis:
            return self._v.copy()
        if idx == ():
            return self._v[()] if self._v.ndim == 0 else self._v.copy()
        return self._v[idx]

    @property
    def shape(self):
        return self._v.shape


class _Group(dict):
    def create_group(self, name):
        g = self[name] = _Group()
        return g

    def create_dataset(self, name, data=None):
        self[name] = _Leaf(data)
        return self[name]


class DictDataset(_Group):
    """{video_key: {field: array}} with the h5py read protocol (+ create_group/create_dataset for predictions)."""
    def __init__(self, videos=None, path=None):
        super().__init__()
        self.filename = path
        for k, fields in (videos or {}).items():
            g = self.create_group(k)
            for f, v in fields.items():
                g.create_dataset(f, data=v)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if getattr(self, "_save_to", None):
            flat = {}
            def walk(prefix, g):
                for k, v in g.items():
                    if isinstance(v, _Group):
                        walk(f"{prefix}{k}/", v)
                    else:
                        flat[f"{prefix}{k}"] = v[...]
            walk("", self)
            np.savez_compressed(self._save_to, **flat)

    @classmethod
    def from_npz(cls, path):
        z = np.load(path, allow_pickle=False)
        videos = {}
        for name in z.files:
            key, field = name.rsplit("/", 1)
            videos.setdefault(key, {})[field] = z[name]
        return cls(videos, path=path)

    def save_npz(self, path):
        self._save_to = path
        self.close()


def open_dataset(path, mode="r"):
    if mode == "r":
        if isinstance(path, DictDataset):
            return path
        if str(path).endswith(".npz"):
            return DictDataset.from_npz(path)
    try:
        import h5py
    except ImportError:
        if mode == "w":                       # predictions without h5py: same group layout, flattened into an .npz
            ds = DictDataset(path=path)
            ds._save_to = path if str(path).endswith(".npz") else str(path) + ".npz"
            return ds
        raise ImportError(f"h5py is required to open {path}; alternatively pass an .npz / DictDataset "
                          "(summarizer_amd.utils.datasets)")
    return h5py.File(path, mode)


def synthetic_dataset(n_videos, seed, D=1024, t_range=(150, 320), n_users=20, key_fmt="video_{}"):
    """SumMe/TVSum-shaped synthetic set: non-negative pool5-like features, block-constant gtscores,
    picks every 15th frame, random change points, binary user summaries at 15% density."""
    rng = np.random.default_rng(seed)
    videos = {}
    for i in range(n_videos):
        T = int(np.ceil(rng.uniform(*t_range)))
        n_frames = int(15 * T - rng.integers(0, 15))
        picks = (15 * np.arange(T)).astype(np.int32)
        n_seg = max(1, T // 15)
        cuts = np.sort(rng.choice(np.arange(15, n_frames - 15), size=n_seg - 1, replace=False)) if n_seg > 1 else np.array([], int)
        starts = np.concatenate([[0], cuts]).astype(np.int64)
        ends = np.concatenate([cuts - 1, [n_frames - 1]]).astype(np.int64)
        gt = rng.random((T + 1) // 2).repeat(2)[:T].astype(np.float32)
        us = np.clip(gt[None, :] + 0.3 * rng.standard_normal((n_users, T)), 0, 1).astype(np.float32)
        usf = np.repeat(us, 15, axis=1)[:, :n_frames]
        if usf.shape[1] < n_frames:
            usf = np.pad(usf, ((0, 0), (0, n_frames - usf.shape[1])))
        # features weakly correlated with the target so that training has something to learn
        base = 0.5 * np.abs(rng.standard_normal((T, D))).astype(np.float32)
        base[:, :16] += gt[:, None]
        videos[key_fmt.format(i + 1)] = dict(
            features=base.astype(np.float32), gtscore=gt, user_scores=usf.astype(np.float32),
            user_summary=(rng.random((n_users, n_frames)) < 0.15).astype(np.float32),
            change_points=np.stack([starts, ends], axis=1).astype(np.int32),
            n_frame_per_seg=(ends - starts + 1).astype(np.int32), n_frames=np.int64(n_frames), n_steps=np.int64(T),
            picks=picks)
    return DictDataset(videos)
