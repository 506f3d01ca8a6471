"""Feature ingest for scoring host-resident videos.

The reference uploads one video at a time from pageable memory and waits for it (`torch.from_numpy(seq).unsqueeze(1).cuda()`,
summarizer/models/__init__.py:47-51) -- 0.5 M frames/s on this node, 17x below what the scorer sustains.  `StreamingScorer`
keeps the GPU fed from host memory instead:
  * videos are grouped into packed batches of up to `max_frames` frames;
  * a batch is packed back to back into a PINNED staging buffer by a pool of native memcpy threads (`sumk_pack_rows`),
    shipped with ONE asynchronous H2D copy on a copy stream, scored with one packed launch on the compute stream, and its
    scores come back through a pinned buffer with one asynchronous D2H copy;
  * `depth` staging slots rotate, so packing + H2D of batch i+1 overlap the scoring of batch i.
Scores are bit-identical to `model.score_packed` on resident features (tests/test_gpu_ingest.py).

Sources: any iterable of (key, float32 (T, D) array), or a feature STORE -- an .npz with "<video>/features" entries, an HDF5 file in
the reference's schema (summarizer/datasets/README.md:5-42; needs h5py), or a dict-backed dataset -- through `score_store`, which
reads one video at a time (the store is never held in memory as a whole).
`stage_dtype="bf16"` (opt-in, LOSSY): the native packer converts fp32 -> bf16 while it copies, so the pinned staging buffer and the
H2D copy carry half the bytes (the PCIe link is the bound of the split-bf16 scoring modes); the device widens the batch back to fp32
and scores bf16(features) -- identical to scoring features that were rounded to bf16 beforehand, NOT to the fp32 features.
"""
import ctypes as C
from collections import deque

import numpy as np
import torch

from . import _lib
from ._lib import SumkError


class _Slot:
    def __init__(self, frames, D, dev, stage_dtype=torch.float32):
        self.capacity, self.stage_dtype = frames, stage_dtype
        self.host_x = torch.empty(frames, D, dtype=stage_dtype).pin_memory()
        self.dev_x = torch.empty(frames, D, dtype=torch.float32, device=dev)
        self.dev_stage = torch.empty(frames, D, dtype=stage_dtype, device=dev) if stage_dtype != torch.float32 else None
        self.host_s = torch.empty(frames, dtype=torch.float32).pin_memory()
        self.h2d_done = torch.cuda.Event()
        self.all_done = torch.cuda.Event()
        self.pending = None          # (keys, lens) of the batch in flight


class StreamingScorer:
    def __init__(self, model, max_frames=16384, depth=2, pack_threads=0, stage_dtype="fp32"):
        if getattr(model, "max_length", None):
            raise SumkError("StreamingScorer packs videos back to back; models with positional embeddings score per video")
        p = next(model.parameters())
        if not p.is_cuda:
            raise SumkError("StreamingScorer needs the model on a GPU (summarizer_amd has no CPU path)")
        self.model, self.dev, self.D = model, p.device, int(model.input_size)
        self.max_frames, self.pack_threads = int(max_frames), int(pack_threads)
        if stage_dtype not in ("fp32", "bf16"):
            raise SumkError(f"StreamingScorer: stage_dtype must be 'fp32' or 'bf16', got {stage_dtype!r}")
        self.stage_dtype = torch.bfloat16 if stage_dtype == "bf16" else torch.float32
        self.slots = [_Slot(self.max_frames, self.D, self.dev, self.stage_dtype) for _ in range(max(1, int(depth)))]
        self.copy_stream = torch.cuda.Stream(self.dev)
        self.compute_stream = torch.cuda.Stream(self.dev)
        self.check_stream = torch.cuda.Stream(self.dev)
        self._lib = _lib.load()

    # ------------------------------------------------------------------ one batch through a slot
    def _submit(self, slot, batch):
        keys = [k for k, _ in batch]
        arrs = []
        for k, a in batch:
            a = np.asarray(a)
            if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] != self.D or a.shape[0] == 0:
                raise SumkError(f"video {k!r}: expected non-empty float32 (T, {self.D}) features, got {a.dtype} {a.shape}")
            arrs.append(np.ascontiguousarray(a))
        lens = [a.shape[0] for a in arrs]
        n = int(sum(lens))
        if n > slot.capacity:                                   # a single video longer than the staging buffers: grow this slot
            slot.__init__(n, self.D, self.dev, self.stage_dtype)
        srcs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        nrows = np.asarray(lens, dtype=np.int32)
        bf16 = self.stage_dtype == torch.bfloat16
        pack = self._lib.sumk_pack_rows_bf16 if bf16 else self._lib.sumk_pack_rows
        _lib.check(pack(C.c_void_p(slot.host_x.data_ptr()), srcs, _lib.host_i32(nrows), len(arrs), self.D, self.pack_threads),
                   "sumk_pack_rows_bf16" if bf16 else "sumk_pack_rows")
        with torch.cuda.stream(self.copy_stream):
            (slot.dev_stage if bf16 else slot.dev_x)[:n].copy_(slot.host_x[:n], non_blocking=True)
            slot.h2d_done.record(self.copy_stream)
        with torch.cuda.stream(self.compute_stream), torch.no_grad():
            self.compute_stream.wait_event(slot.h2d_done)
            if bf16:                                            # widen on the device (exact): the scorer's operands are fp32
                from . import kernels
                kernels.cast_bf16_f32(slot.dev_stage[:n], slot.dev_x[:n])
            scores = self.model.score_packed(slot.dev_x[:n], lens)
            slot.host_s[:n].copy_(scores, non_blocking=True)
            slot.all_done.record(self.compute_stream)
        slot.pending = (keys, lens)

    def _collect(self, slot):
        keys, lens = slot.pending
        slot.pending = None
        slot.all_done.synchronize()
        # this batch is home: did a persistent recurrence kernel time out?  (own stream: synchronising the compute stream here
        # would also wait for the batches still in flight behind this one)
        _lib.check(self._lib.sumk_health_check(C.c_void_p(self.check_stream.cuda_stream)), "sumk_health_check")
        flat = slot.host_s[:sum(lens)].numpy()
        off = np.concatenate([[0], np.cumsum(lens)])
        return [(k, flat[off[i]:off[i + 1]].copy()) for i, k in enumerate(keys)]

    # ------------------------------------------------------------------ public
    def score(self, videos):
        """videos: iterable of (key, float32 (T, D) array).  Yields (key, float32 (T,) scores) in input order."""
        was_training = self.model.training
        self.model.eval()
        # entry ordering: whatever the caller queued on ITS stream (model.to(dev), load_state_dict, the last optimiser step)
        # must be visible to the scoring kernels that read the weights on the private compute stream
        self.compute_stream.wait_stream(torch.cuda.current_stream(self.dev))
        try:
            inflight = deque()
            batch, frames, turn = [], 0, 0

            def flush():
                nonlocal batch, frames, turn
                slot = self.slots[turn % len(self.slots)]
                turn += 1
                out = []
                if slot.pending is not None:                    # the slot's previous batch must be home before its buffers are reused
                    assert inflight and inflight[0] is slot
                    out = self._collect(inflight.popleft())
                self._submit(slot, batch)
                inflight.append(slot)
                batch, frames = [], 0
                return out

            for key, feats in videos:
                T = int(np.shape(feats)[0])
                if batch and frames + T > self.max_frames:
                    yield from flush()
                batch.append((key, feats)); frames += T
            if batch:
                yield from flush()
            while inflight:
                yield from self._collect(inflight.popleft())
        finally:
            self.model.train(was_training)
            torch.cuda.current_stream(self.dev).wait_stream(self.compute_stream)

    def score_dict(self, videos):
        return dict(self.score(videos.items() if hasattr(videos, "items") else videos))

    def score_store(self, store, keys=None, field="features"):
        """Scores the videos of a feature STORE, reading one video at a time: `store` = path of an .npz whose entries are named
        "<video>/<field>" (DictDataset.save_npz), path of an HDF5 file in the reference's schema (h5py required), or any object with
        the h5py mapping protocol (`store[key][field][...]`).  Yields (key, scores) in `keys` order (default: the store's order)."""
        yield from self.score(iter_store(store, keys, field))


def iter_store(store, keys=None, field="features"):
    """(key, float32 (T, D) array) for every requested video of a feature store, one read per video (models/__init__.py:47-51 reads
    `dataset[key]["features"][...]` the same way)."""
    import os
    if isinstance(store, (str, os.PathLike)) and str(store).endswith(".npz"):
        with np.load(store, allow_pickle=False) as z:           # NpzFile: members are read (and decompressed) on access only
            names = {n.rsplit("/", 1)[0]: n for n in z.files if n.endswith("/" + field)}
            for k in (keys if keys is not None else names):
                if k not in names:
                    raise KeyError(f"{store}: no '{k}/{field}' entry")
                yield k, np.ascontiguousarray(z[names[k]], dtype=np.float32)
        return
    if isinstance(store, (str, os.PathLike)):
        from .utils.datasets import open_dataset
        store = open_dataset(store, "r")                        # HDF5 (h5py) -- raises ImportError with a pointer to .npz otherwise
    for k in (keys if keys is not None else list(store.keys())):
        yield k, np.ascontiguousarray(store[k][field][...], dtype=np.float32)
