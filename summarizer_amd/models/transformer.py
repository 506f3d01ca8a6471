"""Transformer-encoder scorer on MI355X -- drop-in for `summarizer/models/transformer.py` (reference): model class and
trainer (the first "next" model after the north_star scorers).

Same constructor (transformer.py:19), same parameter names / state_dict keys (the stock `nn.TransformerEncoder` objects are
kept as PARAMETER CONTAINERS and never called), same forward contract x (seq_len, batch, input_size) -> (seq_len, batch, 1).
Quirks reproduced: one `layer_norm` used as the encoder's final norm AND after k1 (transformer.py:47,50,100); the in-place
positional add with the sinusoid table's batch re-view (transformer.py:83-89).  Training runs through HIP backward kernels;
its dropouts (0.1 inside the encoder layers, 0.5 after k1) use the deterministic hash masks of the VASNet path.
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
from .vasnet import _sinusoid_table
from ..training import FlatAdam, dist_info, plan_shards, step_video_total


def _k_one(loss):
    return kernels.one(loss.device) if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32 else None


class Transformer(nn.Module):
    def __init__(self, input_size=1024, encoder_layers=6, attention_heads=8, more_residuals=False, max_length=None,
                 pos_embed="simple", epsilon=1e-5, weight_init=None):
        super().__init__()
        self.input_size = input_size
        self.encoder_layers, self.attention_heads, self.epsilon = encoder_layers, attention_heads, epsilon
        self.max_length = max_length
        if self.max_length:
            self.pos_embed_type = pos_embed
            if pos_embed == "simple":
                self.pos_embed = nn.Embedding(self.max_length, self.input_size)
            elif pos_embed == "attention":
                self.pos_embed = _sinusoid_table(self.max_length, self.input_size)
            else:
                self.max_length = None
        self.more_residuals = more_residuals
        self.dropout = nn.Dropout(0.5)
        self.layer_norm = nn.LayerNorm(self.input_size, epsilon)
        # creation order and objects as in transformer.py:49-53 (so seeds and state_dict keys line up)
        self.transformer_encoder_layer = nn.TransformerEncoderLayer(d_model=input_size, nhead=attention_heads,
                                                                    dim_feedforward=self.input_size, dropout=0.1,
                                                                    activation="relu")
        self.transformer_encoder = nn.TransformerEncoder(encoder_layer=self.transformer_encoder_layer,
                                                         num_layers=encoder_layers, norm=self.layer_norm,
                                                         enable_nested_tensor=False)
        self.k1 = nn.Linear(self.input_size, self.input_size)
        self.k2 = nn.Linear(self.input_size, 1)
        if weight_init:
            fn = {"he": init.kaiming_uniform_, "kaiming": init.kaiming_uniform_, "xavier": init.xavier_uniform_}.get(weight_init.lower())
            if fn is not None:                                                  # transformer.py:58-70
                for i in range(self.transformer_encoder.num_layers):
                    fn(self.transformer_encoder.layers[i].linear1.weight)
                    fn(self.transformer_encoder.layers[i].linear2.weight)
                fn(self.k1.weight); fn(self.k2.weight)
        self._pos_rows_cache = {}
        self._seed_counter = 0

    def _pos(self, T, B, device):
        if self.max_length is None:
            return None, None
        assert self.max_length >= T, "input sequence has higher length than max_length"
        key = (T, B, str(device))
        rows = self._pos_rows_cache.get(key)
        if rows is None:
            r = np.arange(B * T)
            idx = (r % T) if self.pos_embed_type == "simple" else (r // B)
            rows = self._pos_rows_cache[key] = torch.from_numpy(idx.astype(np.int32)).to(device)
        if self.pos_embed_type == "simple":
            return self.pos_embed.weight, rows
        if self.pos_embed.device != device:
            self.pos_embed = self.pos_embed.to(device)
        return self.pos_embed, rows

    def forward(self, x):
        """x: (seq_len, batch_size, input_size) -> (seq_len, batch_size, 1)"""
        seq_len, batch_size, input_size = x.shape
        kernels._require_gpu(x, "Transformer.forward")
        if batch_size == 1 and x.is_contiguous():
            xp = x.view(seq_len, input_size)
        else:
            xp = x.permute(1, 0, 2).contiguous().view(batch_size * seq_len, input_size)
        table, rows = self._pos(seq_len, batch_size, x.device)
        s = self._score(xp, kernels.SeqBatch.get([seq_len] * batch_size, x.device), table, rows)
        if table is not None and xp.data_ptr() != x.data_ptr():
            with torch.no_grad():
                x.copy_(xp.view(batch_size, seq_len, input_size).permute(1, 0, 2))
        return s.view(batch_size, seq_len, 1).permute(1, 0, 2)

    def score_packed(self, x_packed, lens):
        assert self.max_length is None, "score_packed does not take positional embeddings (use forward)"
        return self._score(x_packed, kernels.SeqBatch.get(lens, x_packed.device), None, None)

    def _score(self, xp, sb, table, rows):
        p = dict(self.named_parameters())
        opts = dict(layer_eps=1e-5, final_eps=self.epsilon, more_residuals=self.more_residuals,
                    precision=getattr(self, "precision", "fp32"))
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            from ..autograd import TransformerFunction
            if self.training:
                self._seed_counter += 1
                opts.update(layer_dropout_p=float(self.transformer_encoder.layers[0].dropout.p),
                            head_dropout_p=float(self.dropout.p),
                            seed=(torch.initial_seed() * 1000003 + self._seed_counter) & (2**63 - 1))
            names = kernels.transformer_param_names(self.encoder_layers)
            cfg = dict(n_layers=self.encoder_layers, n_heads=self.attention_heads, dff=self.input_size)
            return TransformerFunction.apply(xp, sb, cfg, opts, table, rows, names, *[p[n] for n in names])
        if opts["precision"] in kernels.PLANES_OF:          # inference in a split-bf16 arithmetic: every projection on the plane GEMM
            opts["wplanes"] = self._wplanes(p, opts["precision"])
        scores, _ = kernels.transformer_forward_packed(xp, sb, p, self.encoder_layers, self.attention_heads, self.input_size,
                                                       opts, table, rows)
        return scores

    def _wplanes(self, p, precision):
        """Cached weight-plane block (kernels.transformer_wplanes) of the current weights.  Keyed like VASNet's: storage addresses and
        tensor versions (what torch can see), kernels.WEIGHTS_EPOCH (optimiser steps through the C ABI), precision and stream."""
        names = kernels.transformer_param_names(self.encoder_layers)
        ps = [p[n] for n in names]
        key = tuple(q.data_ptr() for q in ps) + tuple(q._version for q in ps) + (
            kernels.WEIGHTS_EPOCH[0], precision, torch.cuda.current_stream(ps[0].device).cuda_stream if ps[0].is_cuda else 0)
        if getattr(self, "_wpl", None) is None or self._wpl_key != key:
            same_stream = getattr(self, "_wpl_stream", None) == key[-1]      # (VASNet._wplanes: no re-use across streams)
            with torch.no_grad():
                self._wpl = kernels.transformer_wplanes({n: q.detach() for n, q in zip(names, ps)}, self.input_size, self.input_size,
                                                        self.encoder_layers, kernels.PLANES_OF[precision], out=getattr(self, "_wpl_buf", None) if same_stream else None)
            self._wpl_buf = getattr(self._wpl, "_sumk_keep", None) if self._wpl is not None else None
            self._wpl_key, self._wpl_stream = key, key[-1]
        return self._wpl


class TransformerTrainer(Trainer):
    """Mirror of the reference trainer (transformer.py:106-192): per-video MSE regression with Adam, periodic test, best-
    correlation weights; same `extra_params`.  Extensions as for VASNetTrainer: `batch_videos`, video sharding + one
    flat-bucket gradient all-reduce per step under torch.distributed."""
    def _init_model(self):
        ep = self.hps.extra_params
        model = Transformer(
            encoder_layers=int(ep.get("encoder_layers", 6)),
            attention_heads=int(ep.get("attention_heads", 8)),
            more_residuals=ep.get("more_residuals", False),
            max_length=int(ep["max_pos"]) if "max_pos" in ep else None,
            pos_embed=ep.get("pos_embed", "simple"),
            epsilon=float(ep.get("epsilon", 1e-5)),
            weight_init=ep.get("weight_init", None),
            **({"input_size": int(ep["input_size"])} if "input_size" in ep else {}))
        model.precision = ep.get("precision", "fp32")      # "fp32" | "bf16x3" (kernels.precision_code)
        if self.hps.use_cuda:
            torch.cuda.set_device(self.hps.cuda_device)
            model.cuda()
        return model

    def train(self, fold):
        self.model.train()
        train_keys, _ = self._get_train_test_keys(fold)
        self.draw_gtscores(fold, train_keys)
        dev = self._device()
        rank, world = dist_info()
        bv = int(self.hps.extra_params.get("batch_videos", 1))
        used = set(kernels.transformer_param_names(self.model.encoder_layers)) | {"pos_embed.weight"}
        self.optimizer = FlatAdam([p for n, p in self.model.named_parameters() if n in used and p.requires_grad],
                                  lr=self.hps.lr, weight_decay=self.hps.weight_decay,
                                  comm_dtype=torch.bfloat16 if getattr(self.model, "precision", "fp32") == "bf16" else None)
        self.optimizer.broadcast()                 # identical weights on every rank: ONE collective over the flat bucket
        my_keys, sizes, steps_per_epoch = plan_shards(train_keys, lambda: [self.dataset[k]["features"].shape[0] for k in train_keys], bv)
        best = self._fold_best()
        packed = self.model.max_length is None
        for epoch in range(self.hps.epochs):
            losses, dist_scores = [], {}
            random.shuffle(my_keys)
            for step in range(steps_per_epoch):
                keys = my_keys[step * bv:(step + 1) * bv]
                self.optimizer.zero_grad()
                if keys:
                    vids = [self._video_on_device(k, dev, want_target=True) for k in keys]
                    if packed:
                        lens_b = [v[0].shape[0] for v in vids]
                        x = vids[0][0] if len(vids) == 1 else torch.cat([v[0] for v in vids])
                        target = vids[0][1] if len(vids) == 1 else torch.cat([v[1] for v in vids])
                        scores = self.model.score_packed(x, lens_b)
                        n_total = len(lens_b) if world == 1 else step_video_total(sizes, bv, step)
                        loss = SegmentMseMeanFunction.apply(scores, target, kernels.SeqBatch.get(lens_b, dev), 1.0 / n_total)   # mean over videos of the MSE per video (transformer.py:161)
                        for k, piece in zip(keys, torch.split(scores.detach(), lens_b)):
                            dist_scores[k] = piece.view(-1, 1, 1)
                    else:
                        loss = 0
                        for k, (seq, target) in zip(keys, vids):
                            sc = self.model(seq.unsqueeze(1).clone())
                            loss = loss + torch.mean((sc.view(-1) - target) ** 2) / (len(vids) if world == 1 else step_video_total(sizes, bv, step))
                            dist_scores[k] = sc.detach()
                    loss.backward(gradient=_k_one(loss))
                    losses.append(loss.detach())
                self.optimizer.step(grad_scale=self.optimizer.all_reduce_grads(average=False))
            train_avg_loss = float(torch.stack(losses).mean()) if losses else float("nan")
            self.log.info(f"Epoch: {f'{epoch+1}/{self.hps.epochs}':6}   Loss: {train_avg_loss:.05f}")
            self.hps.writer.add_scalar(f"{self.dataset_name}/Fold_{fold+1}/Train/Loss", train_avg_loss, epoch)
            self._evaluate_epoch(fold, epoch, best)
        self.draw_scores(fold, dist_scores)
        return best[0], best[1], best[2]
