"""VASNet scorer on MI355X -- drop-in for `summarizer/models/vasnet.py` (reference) at the model-class level.

Same constructor signature, same parameter names / state_dict keys, same forward contract
`x (seq_len, batch, input_size) -> (seq_len, batch, 1)` (vasnet.py:18, 92-98), but the arithmetic runs in
libsumk.so (HIP, gfx950) through `summarizer_amd.kernels`.  Reference quirks are reproduced, not fixed:
the single LayerNorm applied twice (vasnet.py:137,143), scale = 1/sqrt(D) (vasnet.py:34), the tril*triu
aperture mask (vasnet.py:126-127), the in-place positional-embedding add that mutates the caller's tensor
(vasnet.py:109,111) including its batch>1 re-view for the sinusoid table.
Extension (not in the reference): `score_packed` scores MANY videos of different lengths in one launch --
that is the path `Trainer.test` / bench.py use to fill the GPU.
"""
import math
import random
import numpy as np
import torch
import torch.nn as nn
import torch.nn.init as init

from .. import kernels
from ..autograd import SegmentMseMeanFunction
from . import Trainer
from ..training import FlatAdam, dist_info, plan_shards, step_video_total


def _k_one(loss):
    return kernels.one(loss.device) if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32 else None


class VASNet(nn.Module):
    def __init__(self, input_size=1024, max_length=None, pos_embed="simple", ignore_self=False,
                 attention_aperture=None, scale=None, epsilon=1e-6, weight_init="xavier", precision="fp32", fold_vo=False):
        super().__init__()
        self.precision = precision      # GEMM arithmetic: "fp32" (exact) | "bf16x6" | "bf16x3" | "bf16" (kernels.precision_code); not in the reference
        # fold_vo (not in the reference, opt-in): inference folds the value and output projections into ONE matrix Wvo = Wo.Wv
        # (recomputed whenever the weights may have changed), which removes the out-projection GEMM: 18 % fewer FLOPs per frame,
        # scores equal up to fp32 re-association (~1e-6).  Training always runs the reference's operation order.
        self.fold_vo = bool(fold_vo)
        self._wvo, self._wvo_key = None, None
        self.input_size = input_size
        self.aperture = attention_aperture
        self.ignore_self = ignore_self
        self.scale = scale if scale is not None else 1 / np.sqrt(self.input_size)
        self.epsilon = epsilon

        # module creation order follows vasnet.py:42-66 so that a given torch seed yields the same weights
        self.max_length = max_length
        if self.max_length:
            self.pos_embed_type = pos_embed
            if pos_embed == "simple":
                self.pos_embed = nn.Embedding(self.max_length, self.input_size)
            elif pos_embed == "attention":
                self.pos_embed = _sinusoid_table(self.max_length, self.input_size)   # plain tensor, like the reference
            else:
                self.max_length = None
        self.dropout = nn.Dropout(0.5)
        self.layer_norm = nn.LayerNorm(self.input_size, epsilon)
        self.K = nn.Linear(self.input_size, self.input_size, bias=False)
        self.Q = nn.Linear(self.input_size, self.input_size, bias=False)
        self.V = nn.Linear(self.input_size, self.input_size, bias=False)
        self.attention_head_projection = nn.Linear(self.input_size, self.input_size, bias=False)
        self.k1 = nn.Linear(self.input_size, self.input_size)
        self.k2 = nn.Linear(self.input_size, 1)

        mats = [self.K, self.Q, self.V, self.attention_head_projection, self.k1, self.k2]   # vasnet.py:70-86
        for m in mats:
            if weight_init.lower() in ["he", "kaiming"]:
                init.kaiming_uniform_(m.weight)
            else:
                init.xavier_uniform_(m.weight, gain=np.sqrt(2.0))
        init.constant_(self.k1.bias, 0.1)
        init.constant_(self.k2.bias, 0.1)
        self._pos_rows_cache = {}
        self._seed_counter = 0

    # ------------------------------------------------------------------ helpers
    def _params(self):
        return {k: v for k, v in self.named_parameters()}

    def _opts(self, training):
        o = dict(scale=float(self.scale), eps=float(self.epsilon), ignore_self=bool(self.ignore_self),
                 aperture=self.aperture, precision=self.precision)
        if self.aperture is not None:
            assert isinstance(self.aperture, int)      # vasnet.py:125
        if training:
            self._seed_counter += 1
            o.update(dropout_p=float(self.dropout.p), seed=(torch.initial_seed() * 1000003 + self._seed_counter) & (2**63 - 1))
            # a step captured into a HIP graph (VASNetTrainer): the kernels add this device word to the seed when they RUN, and the
            # captured step increments it, so every replay draws fresh dropout masks (kernel arguments are frozen at capture)
            if getattr(self, "graph_seed", None) is not None:
                o["seed_dev"] = self.graph_seed
        if getattr(self, "tail_grads_ready_event", None) is not None:
            o["tail_grads_ready_event"] = self.tail_grads_ready_event     # data-parallel trainers: see VASNetTrainer.train
        return o

    def _pos(self, T, B, device):
        """(table, rows) for the in-place positional add over batch-major packed rows r = b*T + t."""
        if self.max_length is None:
            return None, None
        assert self.max_length >= T, "input sequence has higher length than max_length"     # vasnet.py:107
        key = (T, B, str(device))
        rows = self._pos_rows_cache.get(key)
        if rows is None:
            r = np.arange(B * T)
            idx = (r % T) if self.pos_embed_type == "simple" else (r // B)     # vasnet.py:108 vs :111 (re-view quirk)
            rows = self._pos_rows_cache[key] = torch.from_numpy(idx.astype(np.int32)).to(device)
        if self.pos_embed_type == "simple":
            table = self.pos_embed.weight
        else:
            if self.pos_embed.device != device:
                self.pos_embed = self.pos_embed.to(device)
            table = self.pos_embed
        return table, rows

    # ------------------------------------------------------------------ reference interface
    def forward(self, x):
        """x: (seq_len, batch_size, input_size) -> (seq_len, batch_size, 1)"""
        seq_len, batch_size, input_size = x.shape
        assert self.input_size == input_size                                   # vasnet.py:104
        kernels._require_gpu(x, "VASNet.forward")
        if batch_size == 1 and x.is_contiguous():
            xp = x.view(seq_len, input_size)                                   # zero-copy; pos add lands in caller's x
        else:
            xp = x.permute(1, 0, 2).contiguous().view(batch_size * seq_len, input_size)
        table, rows = self._pos(seq_len, batch_size, x.device)
        sb = kernels.SeqBatch.get([seq_len] * batch_size, x.device)
        s = self._score(xp, sb, table, rows)
        if table is not None and xp.data_ptr() != x.data_ptr():
            with torch.no_grad():                                              # mirror the caller-visible mutation
                x.copy_(xp.view(batch_size, seq_len, input_size).permute(1, 0, 2))
        return s.view(batch_size, seq_len, 1).permute(1, 0, 2)

    # ------------------------------------------------------------------ batched extension
    def score_packed(self, x_packed, lens):
        """x_packed: (sum(lens), D) frames of several videos back to back -> (sum(lens),) scores."""
        assert self.max_length is None, "score_packed does not take positional embeddings (use forward)"
        sb = kernels.SeqBatch.get(lens, x_packed.device)
        return self._score(x_packed, sb, None, None)

    def _score(self, xp, sb, table, rows):
        training = self.training and torch.is_grad_enabled()
        if training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            from ..autograd import VasnetFunction
            names = [k for _, k in kernels.VASNET_FIELDS]
            p = self._params()
            return VasnetFunction.apply(xp, sb, self._opts(self.training), table, rows, names, *[p[n] for n in names])
        opts = self._opts(False)
        wvo = self._folded() if self.fold_vo else None
        if self.precision in kernels.PLANES_OF and table is None and self.input_size % 256 == 0 and sb.n_rows >= 256:
            opts["wplanes"] = self._wplanes(wvo)          # split-bf16 scoring on operand planes (csrc/gemm_pw.hip)
        scores, _ = kernels.vasnet_forward_packed(xp, sb, self._params(), opts, table, rows, training=False, wvo=wvo)
        return scores

    def _weights_key(self):
        """What a cache derived from the weights must be keyed by: storage addresses and tensor versions (what torch can see) plus
        kernels.WEIGHTS_EPOCH (bumped by every optimiser step through the C ABI, invisible to torch)."""
        ps = [p for _, p in sorted(self._params().items())]
        # (+ the current stream: the block is built by kernels on it, a call on another stream builds its own)
        return tuple(p.data_ptr() for p in ps) + tuple(p._version for p in ps) + (kernels.WEIGHTS_EPOCH[0], self.precision, bool(self.fold_vo),
                                                                                  torch.cuda.current_stream(ps[0].device).cuda_stream if ps[0].is_cuda else 0)

    def _wplanes(self, wvo):
        """Cached weight-plane block (kernels.vasnet_wplanes) of the current weights; rebuilt on any weight change (same rules as
        _folded; invalidate_folded() drops it too)."""
        key = self._weights_key()
        if getattr(self, "_wpl", None) is None or self._wpl_key != key:
            # the backing block is re-used only by the stream that built it: kernels another stream has queued may still read the old planes
            same_stream = getattr(self, "_wpl_stream", None) == key[-1]
            with torch.no_grad():
                self._wpl = kernels.vasnet_wplanes({k: v.detach() for k, v in self._params().items()}, self.input_size,
                                                   kernels.PLANES_OF[self.precision], wvo=wvo, out=getattr(self, "_wpl_buf", None) if same_stream else None)
            self._wpl_buf = getattr(self._wpl, "_sumk_keep", None) if self._wpl is not None else None
            self._wpl_key, self._wpl_stream = key, key[-1]
        return self._wpl

    def _folded(self):
        """Cached Wvo = Wo.Wv.  The key holds what torch can see (storage addresses and tensor versions) plus
        kernels.WEIGHTS_EPOCH, which every optimiser step through the C ABI bumps (those writes are invisible to torch: a model kept
        in eval() while FlatAdam steps it would otherwise score with a stale Wvo); train() / eval() switches and load_state_dict()
        drop the cache as well.  (Replays of a HIP-graph-captured training step bypass the Python wrappers: call
        invalidate_folded() after them.)"""
        wo, wv = self.attention_head_projection.weight, self.V.weight
        key = (wo.data_ptr(), wv.data_ptr(), wo._version, wv._version, kernels.WEIGHTS_EPOCH[0])
        if self._wvo is None or self._wvo_key != key:
            with torch.no_grad():
                self._wvo = kernels.fold_vo(wo.detach(), wv.detach(), out=self._wvo if self._wvo is not None and self._wvo.device == wo.device else None)
            self._wvo_key = key
        return self._wvo

    def invalidate_folded(self):
        self._wvo_key = None
        self._wpl_key = None

    def train(self, mode=True):
        self._wvo_key = None
        self._wpl_key = None
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self._wvo_key = None
        self._wpl_key = None
        return super().load_state_dict(*args, **kwargs)


def _sinusoid_table(max_length, d):
    """vasnet.py:44-48 (vectorised; same float64 -> float32 rounding as the reference's element-wise loop)."""
    pos = np.arange(max_length, dtype=np.float64)[:, None]
    i = np.arange(0, d, 2, dtype=np.float64)[None, :]
    tab = np.zeros((max_length, d), dtype=np.float32)
    tab[:, 0::2] = np.sin(pos / (10000 ** ((2 * i) / d)))
    tab[:, 1::2] = np.cos(pos / (10000 ** ((2 * (i + 1)) / d)))
    return torch.from_numpy(tab)


class _NotResident(RuntimeError):
    """_capture_step: the video's tensors are not kept in the HBM cache, so a captured step could not keep them alive."""


class VASNetTrainer(Trainer):
    """Mirror of the reference trainer (vasnet.py:151-238): per-video MSE regression to the min-max normalised gtscore,
    Adam(lr, weight_decay), periodic test, best-correlation weights.  Same `extra_params` parsing, quirks included
    (bool("False") is True, vasnet.py:156).

    Extensions (all default to the reference schedule): extra_params["batch_videos"] = videos per optimiser step on each
    rank (default 1); under torch.distributed the training videos are sharded over ranks (balanced by frames) and the
    flat gradient bucket is all-reduced once per step (RCCL).

    The reference schedule itself (one video per optimiser step, vasnet.py:193-212; single process) runs as HIP GRAPHS from the
    second epoch on: the whole step of a video -- zero the gradient bucket, the small-batch forward (csrc/gemm_lean.hip SK
    launches), per-video MSE, backward, fused Adam -- is captured once per video (its features and target stay at fixed addresses in
    the HBM cache) and replayed every later epoch, so the ~45 kernels of a step are not paced by the host (the eager step spends a
    third of its time waiting for Python between the forward and the backward).  Dropout masks stay fresh per replay through a
    device-side seed word (sumk_vasnet_opts::seed_dev).  extra_params["hip_graph"] = "0" keeps every step eager."""

    def _init_model(self):
        ep = self.hps.extra_params
        model = VASNet(
            max_length=int(ep["max_pos"]) if "max_pos" in ep else None,
            pos_embed=ep.get("pos_embed", "simple"),
            ignore_self=bool(ep.get("ignore_self", False)),
            attention_aperture=int(ep["local"]) if "local" in ep else None,
            scale=float(ep["scale"]) if "scale" in ep else None,
            epsilon=float(ep.get("epsilon", 1e-6)),
            weight_init=ep.get("weight_init", "xavier"),
            precision=ep.get("precision", "fp32"),
            fold_vo=bool(ep.get("fold_vo", False)),
            **({"input_size": int(ep["input_size"])} if "input_size" in ep else {}))
        if self.hps.use_cuda:
            self.log.info(f"Setting CUDA device: {self.hps.cuda_device}")
            torch.cuda.set_device(self.hps.cuda_device)
            model.cuda()
        return model

    def _load_video(self, key, dev):
        return self._video_on_device(key, dev, want_target=True)

    def train(self, fold):
        self.model.train()
        train_keys, _ = self._get_train_test_keys(fold)
        self.draw_gtscores(fold, train_keys)
        dev = self._device()
        rank, world = dist_info()
        from ..training import resolve_batch_videos
        bv = resolve_batch_videos(self.hps.extra_params, "vasnet", "bf16" if self.model.precision == "bf16" else "fp32", train_keys, self.log)
        # precision "bf16" = mixed-precision training (BASELINE config 2): bf16 matrix arithmetic with fp32 accumulation, fp32
        # master weights / moments in the flat bucket, and the gradient bucket crossing the all-reduce as bf16
        self.optimizer = FlatAdam(filter(lambda p: p.requires_grad, self.model.parameters()), lr=self.hps.lr,
                                  weight_decay=self.hps.weight_decay,
                                  comm_dtype=torch.bfloat16 if self.model.precision == "bf16" else None)
        self.optimizer.broadcast()                 # identical weights on every rank: ONE collective over the flat bucket
        my_keys, sizes, steps_per_epoch = plan_shards(train_keys, lambda: [self.dataset[k]["features"].shape[0] for k in train_keys], bv)
        # data-parallel overlap: the HIP backward records this event once the gradients of Wo / k1 / k2 (the tail of the
        # bucket) are final; their all-reduce then runs on a side stream under the attention backward + QKV weight gradients
        tail_from = self.optimizer.tail_offset(self.model.attention_head_projection.weight) if world > 1 else None
        self.model.tail_grads_ready_event = torch.cuda.Event() if world > 1 else None

        best = self._fold_best()
        use_packed = self.model.max_length is None
        # reference schedule as HIP graphs: one captured step per video, replayed from the second epoch on (class docstring)
        use_graph = (world == 1 and bv == 1 and use_packed and dev.type == "cuda"
                     and str(self.hps.extra_params.get("hip_graph", "1")) not in ("0", "False", "false"))
        graphs, graph_pool, graphs_zeroed = {}, None, False
        if use_graph:
            self.model.graph_seed = torch.zeros(1, dtype=torch.int64, device=dev)
        for epoch in range(self.hps.epochs):
            losses, dist_scores = [], {}
            random.shuffle(my_keys)
            for step in range(steps_per_epoch):
                keys = my_keys[step * bv:(step + 1) * bv]
                if use_graph and epoch >= 1 and keys:
                    ent = graphs.get(keys[0])
                    if ent is None:
                        try:
                            ent = graphs[keys[0]] = self._capture_step(keys[0], dev, graph_pool)
                            graph_pool = ent[0].pool()
                        except _NotResident:        # a video beyond the HBM cache: THIS key steps eagerly from now on (not retried), the others keep their graphs
                            ent = graphs[keys[0]] = False
                        except Exception as e:      # noqa: BLE001  (a configuration that does not capture keeps the eager loop)
                            self.log.warning(f"HIP graph capture failed ({type(e).__name__}: {e}); training continues eagerly")
                            use_graph, self.model.graph_seed = False, None
                            torch.cuda.synchronize(dev)      # leave no half-finished capture work behind the eager steps
                    if use_graph and ent:
                        if not graphs_zeroed:          # the captured steps keep the gradient bucket zero between them (Adam kernel); the
                            self.optimizer.zero_grad(zeroed_by_step=True); graphs_zeroed = True      # first replay after eager steps starts from an explicit one
                        ent[0].replay()
                        kernels.WEIGHTS_EPOCH[0] += 1      # (the replayed Adam kernel changed the weights behind the Python wrappers: weight-derived caches follow)
                        losses.append(ent[1]); dist_scores[keys[0]] = ent[2]
                        continue
                self.optimizer.zero_grad(zeroed_by_step=True)
                if keys:
                    vids = [self._load_video(k, dev) for k in keys]
                    if use_packed:
                        lens_b = [v[0].shape[0] for v in vids]
                        scores = self.model.score_packed(torch.cat([v[0] for v in vids]) if len(vids) > 1 else vids[0][0], lens_b)
                        off = np.concatenate([[0], np.cumsum(lens_b)])
                        # mean over videos of the per-video MSE (== nn.MSELoss per video, vasnet.py:209, when bv == 1)
                        target = torch.cat([v[1] for v in vids]) if len(vids) > 1 else vids[0][1]
                        # world == 1: plain mean; data-parallel: every video of the GLOBAL step weighs 1/n_total (ranks whose
                        # shard has run out contribute nothing and the divisor shrinks with them)
                        n_total = len(lens_b) if world == 1 else step_video_total(sizes, bv, step)
                        loss = SegmentMseMeanFunction.apply(scores, target, kernels.SeqBatch.get(lens_b, dev), 1.0 / n_total)
                        for i, k in enumerate(keys):
                            dist_scores[k] = scores[off[i]:off[i + 1]].detach().view(-1, 1, 1)
                    else:
                        loss = 0
                        for k, (seq, target) in zip(keys, vids):
                            sc = self.model(seq.unsqueeze(1).clone())   # clone: the positional add is in place, seq is the HBM-cached copy
                            loss = loss + torch.mean((sc.view(-1) - target) ** 2) / (len(vids) if world == 1 else step_video_total(sizes, bv, step))
                            dist_scores[k] = sc.detach()
                    loss.backward(gradient=_k_one(loss))
                    losses.append(loss.detach())
                if tail_from is not None and use_packed:          # every rank issues the same two collectives, videos or not
                    if not keys:
                        self.model.tail_grads_ready_event.record()
                    self.optimizer.reduce_tail_async(tail_from, self.model.tail_grads_ready_event)
                scale = self.optimizer.all_reduce_grads(average=False)
                self.optimizer.step(grad_scale=scale, zero_grad=True)

            train_avg_loss = float(torch.stack(losses).mean()) if losses else float("nan")   # one D2H sync per epoch
            self.log.info(f"Epoch: {f'{epoch+1}/{self.hps.epochs}':6}   Loss: {train_avg_loss:.05f}")
            self.hps.writer.add_scalar(f"{self.dataset_name}/Fold_{fold+1}/Train/Loss", train_avg_loss, epoch)

            self._evaluate_epoch(fold, epoch, best)

        self.draw_scores(fold, dist_scores)
        self.model.graph_seed = None
        return best[0], best[1], best[2]

    def _single_video_step(self, key, dev, grads_are_zero=False, video=None):
        """One optimiser step on one video (vasnet.py:193-212): (loss, scores) as detached tensors.  grads_are_zero: the captured
        form -- the previous step's Adam kernel left the gradient bucket zero and this one does the same, so no fill kernel runs.
        video: the (features, target) device tensors to use instead of loading `key` (a capture passes the tensors it keeps alive)."""
        seq, target = video if video is not None else self._load_video(key, dev)
        lens_b = [seq.shape[0]]
        if not grads_are_zero:
            self.optimizer.zero_grad(zeroed_by_step=True)
        scores = self.model.score_packed(seq, lens_b)
        loss = SegmentMseMeanFunction.apply(scores, target, kernels.SeqBatch.get(lens_b, dev), 1.0)    # one video: the mean over videos is the value itself
        loss.backward(gradient=_k_one(loss))
        self.optimizer.step(grad_scale=1.0, zero_grad=grads_are_zero)
        if self.model.graph_seed is not None:
            # next replay: other dropout masks.  The kernels add this word to the seed captured with the step, and the captured seeds of
            # the per-video graphs are consecutive integers: a stride of 1 made video A's replay g draw the masks of video B's replay
            # g - 1.  A large odd stride keeps (captured seed + word) distinct over videos and replays.
            self.model.graph_seed.add_(0x9E3779B97F4A7C15 - (1 << 64))
        return loss.detach(), scores.detach().view(-1, 1, 1)

    def _capture_step(self, key, dev, pool):
        """(graph, loss, scores): the step of `key` captured into a HIP graph; loss / scores are the graph's static outputs (rewritten
        by every replay).  All captured steps share one memory pool -- they never run concurrently."""
        # The graph records raw addresses of the features and the target: they must be the HBM-cached tensors (alive as long as the
        # trainer) and are ALSO kept in the graph entry -- a video the cache declined (over its byte limit) would be re-uploaded from a
        # pinned buffer INSIDE the capture and every replay would copy from recycled host memory: such a video stays eager.
        if (key, str(dev)) not in self._hbm:
            self._load_video(key, dev)
        if (key, str(dev)) not in self._hbm:
            raise _NotResident(f"video {key} is not resident in the HBM cache")
        video = self._hbm[(key, str(dev))]
        seq = video[0]
        # the outputs live OUTSIDE the shared graph pool (allocated before the capture, written by a copy inside it): tensors a capture
        # leaves in the pool were seen to alias those of later captures into the same pool -- the epoch's mean loss then read eight
        # copies of the last step's loss while the weights were exactly right
        loss_out = torch.zeros((), dtype=torch.float32, device=dev)
        scores_out = torch.empty(seq.shape[0], 1, 1, dtype=torch.float32, device=dev)
        kernels.one(dev)                                   # (the cached root gradient of loss.backward exists before the capture, outside the pool)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool):
            loss, scores = self._single_video_step(key, dev, grads_are_zero=True, video=video)
            loss_out.copy_(loss); scores_out.copy_(scores)
        # (the graph holds raw addresses of the batch descriptor's device arrays -- sequence offsets, prebuilt problem tables: keep the
        #  SeqBatch alive beside it, whatever happens to kernels.SeqBatch's cache)
        return g, loss_out, scores_out, kernels.SeqBatch.get([seq.shape[0]], dev), video
