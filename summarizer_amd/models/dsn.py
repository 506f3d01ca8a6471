"""DSN scorer (bidirectional LSTM + per-frame head) on MI355X -- drop-in for `summarizer/models/dsn.py`.

Same constructor (dsn.py:19), same state_dict keys (`rnn.weight_ih_l0[_reverse]`, ..., `out.0.weight`,
`out.0.bias`), same forward contract x (seq_len, batch, input_size) -> (seq_len, batch, 1) (dsn.py:38-47).
"""
import math
import random
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Bernoulli

from .. import kernels
from ..autograd import PolicyLossFunction
from .._lib import SumkError
from . import Trainer
from ._bilstm import pack_time_major, bilstm_scores, bigru_scores
from ..training import FlatAdam, dist_info, plan_shards, step_video_total


def _k_one(loss):
    return kernels.one(loss.device) if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32 else None


class DSN(nn.Module):
    """Deep Summarization Network"""
    def __init__(self, input_size=1024, hidden_size=256, num_layers=1, cell="lstm"):
        super().__init__()
        assert cell in ["lstm", "gru"], "cell must be either 'lstm' or 'gru'"          # dsn.py:21
        self.cell = cell
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.precision = "fp32"          # GEMM arithmetic: "fp32" (exact) | "bf16x3" (kernels.precision_code); not in the reference
        # parameter containers only (same names / shapes / init as the reference); never called.  The optional GRU cell
        # (dsn.py:28-33) runs on a functional step-by-step path (models/_bilstm.py: GruLayerFunction, csrc/gru.hip) -- the
        # reference's own DSNTrainer always builds DSN(), i.e. the LSTM, which has the persistent kernels.
        rnn = nn.LSTM if cell == "lstm" else nn.GRU
        self.rnn = rnn(input_size, hidden_size, num_layers=num_layers, bidirectional=True)
        self.out = nn.Sequential(nn.Linear(hidden_size * 2, 1), nn.Sigmoid())

    def forward(self, x):
        """x: (seq_len, batch_size, input_size) -> probs (seq_len, batch_size, 1)"""
        T, B, F = x.shape
        kernels._require_gpu(x, "DSN.forward")
        xp, lens = pack_time_major(x)
        s = self.score_packed(xp, lens)
        return s.view(B, T, 1).permute(1, 0, 2)

    def score_packed(self, x_packed, lens):
        """Batched extension: frames of several videos back to back (sum(lens), D) -> (sum(lens),) probabilities."""
        sb = kernels.SeqBatch.get(lens, x_packed.device)
        if self.cell == "gru":
            kernels._require_gpu(x_packed, "DSN.score_packed")
            return bigru_scores(self, x_packed, sb, self.num_layers, self.hidden_size, "out.0.weight", "out.0.bias")
        return bilstm_scores(self, x_packed, sb, "rnn.", self.num_layers, self.hidden_size, "out.0.weight", "out.0.bias")


class DSNTrainer(Trainer):
    """Mirror of the reference trainer (dsn.py:50-236): REINFORCE with `num_episodes` Bernoulli episodes per video,
    diversity-representativeness reward, per-video moving-average baseline, grad-norm clip 5.0, Adam; optional
    supervised BCE term (`sup`).  `extra_params` parsing keeps the reference's quirks: beta = int(...) so the default
    0.01 becomes 0 (dsn.py:52), bool("False") is True.

    Differences underneath: the reward of all episodes comes from ONE HIP call (one Gram GEMM + masked reductions,
    csrc/reward.hip) instead of 2 matmuls per episode; the optimiser is the flat-bucket HIP Adam with the clip folded in.
    Extensions: extra_params["batch_videos"], and sharding of the training videos over torch.distributed ranks with one
    gradient all-reduce per step; per-video baselines stay on the rank that owns the video (static assignment)."""

    def _init_model(self):
        ep = self.hps.extra_params
        self.beta = int(ep.get("beta", 0.01))
        self.num_episodes = int(ep.get("num_episodes", 5))
        self.eps = float(ep.get("eps", 0.5))
        self.far_sim = bool(ep.get("far_sim", False))
        self.temp_dist_thre = int(ep.get("temp_dist_thre", 20))
        self.sup = bool(ep.get("sup", False))
        model = DSN(**({"input_size": int(ep["input_size"])} if "input_size" in ep else {}),
                    **({"hidden_size": int(ep["hidden_size"])} if "hidden_size" in ep else {}))
        model.precision = ep.get("precision", "fp32")
        return model

    def compute_reward(self, seq, actions, far_sim=False, temp_dist_thre=20):
        """Reference signature (dsn.py:185): seq (seq_len,1,input_size), actions (seq_len,1,1) -> 0-d reward tensor."""
        T = seq.shape[0]
        if int(actions.detach().sum().item()) == 1:
            # exactly one picked frame: the reference indexes its (T, T) distance matrix with a 0-dim tensor and `.min(1, ...)` then
            # fails (dsn.py:229-230) -- the same exception type here.  (The batched trainer path below does not go through this
            # helper: there the kernel defines the one-pick reward as (0 + exp(-mean d^2)) / 2 instead of ending the run.)
            raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")
        sb = kernels.SeqBatch.get([T], seq.device)
        r = kernels.dsn_reward(seq.detach().reshape(T, -1).contiguous(), sb, actions.detach().reshape(1, T).contiguous(),
                               far_sim=far_sim, temp_dist_thre=temp_dist_thre)
        return r.reshape(())

    def _load_video(self, key, dev):
        return self._video_on_device(key, dev, want_target=True)

    def _sample_actions(self, dist, n_episodes, keys):
        """(n_episodes, n_rows) 0/1 draws from the frame-selection distribution (dsn.py:125, one `dist.sample()` per
        episode there).  A separate method so tests can replay the draws the reference's torch-CPU generator made."""
        return dist.sample((n_episodes,))

    def train(self, fold):
        self.model.train()
        train_keys, _ = self._get_train_test_keys(fold)
        self.draw_gtscores(fold, train_keys)
        self.log.debug("Parameters: {}".format(sum([_.numel() for _ in self.model.parameters()])))
        dev = self._device()
        rank, world = dist_info()
        from ..training import resolve_batch_videos
        bv = resolve_batch_videos(self.hps.extra_params, "dsn", "fp32", train_keys, self.log)
        self.optimizer = FlatAdam(self.model.parameters(), lr=self.hps.lr, weight_decay=self.hps.weight_decay,
                                  comm_dtype=torch.bfloat16 if getattr(self.model, "precision", "fp32") == "bf16" else None)
        self.optimizer.broadcast()                 # identical weights on every rank: ONE collective over the flat bucket
        my_keys, sizes, steps_per_epoch = plan_shards(train_keys, lambda: [self.dataset[k]["features"].shape[0] for k in train_keys], bv)
        # data-parallel overlap (one BiLSTM layer): the library records this event once the biases' and the reverse direction's gradients are
        # final; the tail of the bucket [reverse direction | head] is then all-reduced on a side stream under the forward direction's
        # weight-gradient GEMMs (sumk_lstm_layer_grads::tail_ready_event)
        tail_from = None
        self.model.tail_grads_ready_event = None
        if world > 1 and getattr(self.model, "num_layers", 1) == 1 and getattr(self.model, "cell", "lstm") == "lstm":
            tail_from = self.optimizer.tail_offset(dict(self.model.named_parameters())["rnn.weight_ih_l0_reverse"])
            self.model.tail_grads_ready_event = torch.cuda.Event()

        # dsn.py:81,84: per-video moving-average baselines and the last reward of every video -- kept ON THE DEVICE (float64 like the
        # reference's Python floats) so that a training step never synchronises with the host: the reference reads E rewards
        # back per video, round 1 of this mirror read one vector per step
        key_index = {key: i for i, key in enumerate(my_keys)}
        baselines = torch.zeros(max(len(my_keys), 1), dtype=torch.float64, device=dev)
        last_reward = torch.full((max(len(my_keys), 1),), float("nan"), dtype=torch.float64, device=dev)
        best = self._fold_best()
        E = self.num_episodes

        for epoch in range(self.hps.epochs):
            losses, dist_scores = [], {}
            random.shuffle(my_keys)
            order = torch.tensor([key_index[k] for k in my_keys], dtype=torch.int64).to(dev) if my_keys else None   # one H2D per epoch
            for step in range(steps_per_epoch):
                keys = my_keys[step * bv:(step + 1) * bv]
                self.optimizer.zero_grad(zeroed_by_step=True)
                if keys:
                    vids = [self._load_video(k, dev) for k in keys]
                    lens_b = [v[0].shape[0] for v in vids]
                    x = torch.cat([v[0] for v in vids]) if len(vids) > 1 else vids[0][0]
                    sb = kernels.SeqBatch.get(lens_b, dev)
                    probs = self.model.score_packed(x, lens_b)            # (sum T,)
                    dist = Bernoulli(probs, validate_args=False)          # (argument validation is a D2H sync; probs come from the sigmoid kernel)
                    actions = self._sample_actions(dist, E, keys)         # (E, sum T)   dsn.py:125
                    rewards = kernels.dsn_reward(x, sb, actions.contiguous(), far_sim=self.far_sim,
                                                 temp_dist_thre=self.temp_dist_thre)       # (E, n_videos)  dsn.py:129-131
                    off = np.concatenate([[0], np.cumsum(lens_b)])
                    idx = order[step * bv:step * bv + len(keys)]
                    base = baselines[idx].float()
                    # all videos of the step at once; for one video this is exactly dsn.py:115-140:
                    #   l_v = [beta (mean p - eps)^2 (+ BCE) - sum_e mean_t log_prob(a_e) (r_e - b)] / E
                    # the length penalty, Bernoulli.log_prob (dsn.py:126), the per-video means and the advantage product are two HIP
                    # kernels (forward / backward: sumk_dsn_policy_loss_*) instead of ~35 element-wise launches and their autograd twins
                    l_v = PolicyLossFunction.apply(probs, sb, actions, rewards, base, self.beta, self.eps)      # (n_videos,)
                    if self.sup:
                        target = torch.cat([v[1] for v in vids]) if len(vids) > 1 else vids[0][1]
                        l_v = l_v + sb.segment_mean(torch.nn.functional.binary_cross_entropy(probs, target, reduction="none")) / float(E)  # dsn.py:117-119,140
                    # data-parallel: every video of the GLOBAL step weighs 1/n_total (see training.step_video_total)
                    loss = l_v.mean() if world == 1 else l_v.sum() / step_video_total(sizes, bv, step)
                    for i, k in enumerate(keys):
                        dist_scores[k] = probs[off[i]:off[i + 1]].detach().view(-1, 1, 1)
                    loss.backward(gradient=_k_one(loss))
                    losses.append(loss.detach())
                    mean_r = rewards.detach().mean(dim=0).double()
                    baselines.index_copy_(0, idx, 0.9 * baselines[idx] + 0.1 * mean_r)     # dsn.py:149
                    last_reward.index_copy_(0, idx, mean_r)
                if tail_from is not None:            # every rank issues the same two collectives, videos or not
                    if not keys:
                        self.model.tail_grads_ready_event.record()
                    self.optimizer.reduce_tail_async(tail_from, self.model.tail_grads_ready_event)
                scale = self.optimizer.all_reduce_grads(average=False)
                self.optimizer.step(grad_scale=scale, max_norm=5.0, zero_grad=True)       # clip_grad_norm_(…, 5.0) dsn.py:145, post all-reduce

            epoch_avg_reward = float(torch.nanmean(last_reward)) if my_keys else float("nan")      # the epoch's host sync
            epoch_avg_loss = float(torch.stack(losses).mean()) if losses else float("nan")
            kernels.health_check()               # the epoch's host sync: did any persistent recurrence kernel time out?
            self.log.info(f"Epoch: {f'{epoch+1}/{self.hps.epochs}':6}   Reward: {epoch_avg_reward:.05f}  Loss: {epoch_avg_loss:.05f}")
            self.hps.writer.add_scalar(f"{self.dataset_name}/Fold_{fold+1}/Train/Reward", epoch_avg_reward, epoch)
            self.hps.writer.add_scalar(f"{self.dataset_name}/Fold_{fold+1}/Train/Loss", epoch_avg_loss, epoch)

            self._evaluate_epoch(fold, epoch, best)

        self.draw_scores(fold, dist_scores)
        self.model.tail_grads_ready_event = None
        return best[0], best[1], best[2]
