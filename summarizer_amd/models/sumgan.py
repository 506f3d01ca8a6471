"""SumGAN's LSTM modules on MI355X (`summarizer/models/sumgan.py`): the sLSTM selector -- the scorer, `SumGAN.forward` is
exactly `s_lstm(x)` (sumgan.py:23-46,251-258) -- and the forward-running stacks of the VAE / GAN side: eLSTM (sumgan.py:48-72),
the step-wise decoder dLSTM (sumgan.py:74-115), VAE, Summarizer, cLSTM / GAN and the SumGAN container (sumgan.py:117-258).  Same constructors, forward signatures and state_dict keys; every module is
differentiable through the HIP backward kernels.  `SumGANTrainer` follows the reference trainer (sumgan.py:261-533): VAE
pre-training, then per video the selector+encoder, decoder and discriminator updates with their three Adam optimisers."""
import random

import torch
import torch.nn as nn

from .. import kernels
from . import Trainer
from ._bilstm import pack_time_major, bilstm_scores, lstm_stack
from ..training import FlatAdam


class sLSTM(nn.Module):
    def __init__(self, input_size=1024, hidden_size=1024, num_layers=2):
        """Selector LSTM"""
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.precision = "fp32"          # GEMM arithmetic: "fp32" (exact) | "bf16x3" (kernels.precision_code); not in the reference
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, bidirectional=True)
        self.out = nn.Linear(hidden_size * 2, 1)
        self.sig = nn.Sigmoid()

    def forward(self, x):
        """x: (seq_len, batch_size, input_size) -> scores (seq_len, batch_size, 1)"""
        T, B, F = x.shape
        kernels._require_gpu(x, "sLSTM.forward")
        xp, lens = pack_time_major(x)
        s = self.score_packed(xp, lens)
        return s.view(B, T, 1).permute(1, 0, 2)

    def score_packed(self, x_packed, lens):
        sb = kernels.SeqBatch.get(lens, x_packed.device)
        return bilstm_scores(self, x_packed, sb, "lstm.", self.num_layers, self.hidden_size, "out.weight", "out.bias")


def _unpack_time_major(rows, T, B):
    """(B*T, F) batch-major packed rows -> (T, B, F)."""
    return rows.view(B, T, -1).permute(1, 0, 2)


class eLSTM(nn.Module):
    def __init__(self, input_size=1024, hidden_size=2048, num_layers=2):
        """Encoder LSTM"""
        super().__init__()
        self.precision = "fp32"
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, bidirectional=False)
        self.mu = nn.Linear(hidden_size, hidden_size)
        self.logvar = nn.Linear(hidden_size, hidden_size)

    def forward(self, x):
        """x: (seq_len, batch_size, input_size) -> (h_mu, h_logvar) each (num_layers, batch_size, hidden_size), c_last"""
        from ..autograd import LinearFunction
        kernels._require_gpu(x, "eLSTM.forward")
        xp, lens = pack_time_major(x)
        _, h_last, c_last = lstm_stack(self.lstm, xp, kernels.SeqBatch.get(lens, x.device), precision=self.precision)
        h_mu = LinearFunction.apply(h_last, self.mu.weight, self.mu.bias, self.precision)
        h_logvar = LinearFunction.apply(h_last, self.logvar.weight, self.logvar.bias, self.precision)
        return (h_mu, h_logvar), c_last


class dLSTM(nn.Module):
    def __init__(self, input_size=1024, hidden_size=2048, num_layers=2):
        """Decoder LSTM"""
        super().__init__()
        self.precision = "fp32"
        self.lstm = nn.LSTM(input_size=hidden_size, hidden_size=hidden_size, num_layers=num_layers, bidirectional=False)
        self.recons = nn.Linear(hidden_size, input_size)

    def forward(self, seq_len, h_0, c_0):
        """Decode the entire sequence (the reference's step-by-step loop, sumgan.py:98-115, as ONE op).
        h_0, c_0: (num_layers, batch_size, hidden_size) -> x_hat: (seq_len, batch_size, input_size)"""
        from ..autograd import LstmDecoderFunction, LinearFunction
        kernels._require_gpu(h_0, "dLSTM.forward")
        B, H, L = h_0.size(1), h_0.size(2), self.lstm.num_layers
        sb = kernels.SeqBatch.get([int(seq_len)] * B, h_0.device)
        params = []
        for l in range(L):
            params += [getattr(self.lstm, f"weight_ih_l{l}"), getattr(self.lstm, f"weight_hh_l{l}"),
                       getattr(self.lstm, f"bias_ih_l{l}"), getattr(self.lstm, f"bias_hh_l{l}")]
        rows = LstmDecoderFunction.apply(sb, H, h_0, c_0, *params)                  # (B*T, H) batch-major, time order
        x_hat = LinearFunction.apply(rows, self.recons.weight, self.recons.bias, self.precision)
        return torch.flip(_unpack_time_major(x_hat, int(seq_len), B), (0,))         # reverse (sumgan.py:114)


class VAE(nn.Module):
    def __init__(self, input_size=1024, hidden_size=2048, num_layers=2):
        """Variational Auto Encoder LSTM"""
        super().__init__()
        self.e_lstm = eLSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        self.d_lstm = dLSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)

    def reparameterize(self, mu, logvar):
        std = torch.exp(0.5 * logvar)
        eps = torch.randn_like(std)
        return mu + eps * std

    def forward(self, x):
        """x: (seq_len, batch_size, input_size) -> x_hat (seq_len, batch_size, input_size), (h_mu, h_logvar)"""
        (h_mu, h_logvar), c = self.e_lstm(x)
        h = self.reparameterize(h_mu, h_logvar)
        x_hat = self.d_lstm(x.size(0), h, c)
        return x_hat, (h_mu, h_logvar)


class Summarizer(nn.Module):
    def __init__(self, input_size=1024, sLSTM_hidden_size=1024, sLSTM_num_layers=2, edLSTM_hidden_size=2048,
                 edLSTM_num_layers=2):
        """Summarizer: Selector (sLSTM) + VAE (eLSTM/dLSTM)."""
        super().__init__()
        self.s_lstm = sLSTM(input_size=input_size, hidden_size=sLSTM_hidden_size, num_layers=sLSTM_num_layers)
        self.vae = VAE(input_size=input_size, hidden_size=edLSTM_hidden_size, num_layers=edLSTM_num_layers)

    def forward(self, x, uniform=False):
        """-> x_hat (seq_len, batch_size, input_size), (h_mu, h_logvar), scores (seq_len, batch_size, 1)"""
        if uniform:
            seq_len, batch_size, _ = x.size()
            scores = torch.rand((seq_len, batch_size, 1)).to(x.device)
        else:
            scores = self.s_lstm(x)
        x_weighted = x * scores
        x_hat, (h_mu, h_logvar) = self.vae(x_weighted)
        return x_hat, (h_mu, h_logvar), scores


class cLSTM(nn.Module):
    def __init__(self, input_size=1024, hidden_size=1024, num_layers=2):
        """Discriminator as a classifier LSTM"""
        super().__init__()
        self.precision = "fp32"
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, bidirectional=False)
        self.out = nn.Sequential(nn.Linear(hidden_size, 1), nn.Sigmoid())

    def forward(self, x):
        """x: (seq_len, batch_size, input_size) -> probs (batch_size, 1), h_last (batch_size, hidden_size)"""
        from ..autograd import FrameHeadFunction
        kernels._require_gpu(x, "cLSTM.forward")
        xp, lens = pack_time_major(x)
        _, h_n, _ = lstm_stack(self.lstm, xp, kernels.SeqBatch.get(lens, x.device), precision=self.precision)
        h_last = h_n[-1]                                   # output[-1] of the top layer (sumgan.py:208)
        probs = FrameHeadFunction.apply(h_last, self.out[0].weight, self.out[0].bias)
        return probs.view(-1, 1), h_last


class GAN(nn.Module):
    def __init__(self, input_size=1024, hidden_size=1024, num_layers=2):
        """GAN: discriminator."""
        super().__init__()
        self.c_lstm = cLSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)

    def forward(self, x):
        return self.c_lstm(x)


class SumGAN(nn.Module):
    def __init__(self, input_size=1024, sLSTM_hidden_size=1024, sLSTM_num_layers=2, edLSTM_hidden_size=2048,
                 edLSTM_num_layers=2, cLSTM_hidden_size=1024, cLSTM_num_layers=2):
        """SumGAN: Summarizer + GAN"""
        super().__init__()
        self.summarizer = Summarizer(input_size=input_size, sLSTM_hidden_size=sLSTM_hidden_size, sLSTM_num_layers=sLSTM_num_layers,
                                     edLSTM_hidden_size=edLSTM_hidden_size, edLSTM_num_layers=edLSTM_num_layers)
        self.gan = GAN(input_size=input_size, hidden_size=cLSTM_hidden_size, num_layers=cLSTM_num_layers)

    def forward(self, x):
        """x: (seq_len, batch_size, input_size) -> scores (seq_len, batch_size, 1)"""
        return self.summarizer.s_lstm(x)

    def score_packed(self, x_packed, lens):
        return self.summarizer.s_lstm.score_packed(x_packed, lens)


class SumGANTrainer(Trainer):
    """Mirror of the reference trainer (sumgan.py:262-533), same `extra_params` (`sigma`, `input_size`, `*_hidden_size`,
    `*_num_layers`, `sup`, `pretrain_vae`, `epoch_noise`).  Per video three updates, each with its own Adam:
      selector + encoder:  ||phi(x) - phi(x_hat)||_2 + KL prior + sparsity (|mean(s) - sigma|, or BCE to gtscore with `sup`)
      decoder:             ||phi(x) - phi(x_hat)||_2 + BCE(D(x_hat), 0.9) + BCE(D(x_hat_p), 0.9)      (x_hat_p: uniform scores)
      discriminator:       BCE(D(x), 0.9) + BCE(D(x_hat), 0.1) + BCE(D(x_hat_p), 0.1)   (inputs multiplied by noise early on)
    where phi / D are the cLSTM's last hidden state / probability.  As in the reference, `zero_grad` only clears the group
    being updated and the gradient-norm clip (5.0) spans EVERY parameter of the model -- gradients left in the other groups
    by earlier backward passes count towards the norm and are rescaled with it.  The optimiser math runs in the HIP Adam
    kernel over one flat bucket per group."""

    def _init_model(self):
        ep = self.hps.extra_params
        self.sigma = float(ep.get("sigma", 0.3))
        self.sup = bool(ep.get("sup", False))
        self.pretrain_vae = int(ep.get("pretrain_vae", 20))
        self.epoch_noise = int(ep.get("epoch_noise", 0.2 * self.hps.epochs))
        sizes = {name: int(ep.get(name, default)) for name, default in
                 (("input_size", 1024), ("sLSTM_hidden_size", 1024), ("sLSTM_num_layers", 2), ("edLSTM_hidden_size", 2048),
                  ("edLSTM_num_layers", 2), ("cLSTM_hidden_size", 1024), ("cLSTM_num_layers", 2))}
        model = SumGAN(**sizes)
        for mod in model.modules():                       # "fp32" | "bf16x6" | "bf16x3" for every input projection / dense layer
            if hasattr(mod, "precision"):
                mod.precision = ep.get("precision", "fp32")
        self.log.debug("Generator params: {}".format(sum(p.numel() for p in model.summarizer.parameters())))
        self.log.debug("Discriminator params: {}".format(sum(p.numel() for p in model.gan.parameters())))
        return model

    # ---- losses (sumgan.py:288-321)
    @staticmethod
    def loss_recons(h_real, h_fake):
        return torch.norm(h_real - h_fake, p=2)

    @staticmethod
    def loss_prior(mu, logvar):
        return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())

    def loss_vae(self, x, x_hat, mu, logvar):
        return self.loss_recons(x, x_hat) + self.loss_prior(mu, logvar)

    @staticmethod
    def _bce(p, label):
        return torch.nn.functional.binary_cross_entropy(p, torch.full_like(p, label))

    # ---- optimiser plumbing
    @staticmethod
    def _clip_all(buckets, max_norm=5.0):
        """torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0): one norm over every bucket, every bucket rescaled."""
        acc = torch.zeros(1, dtype=torch.float32, device=buckets[0].flat_grad.device)
        for b in buckets:
            kernels.sumsq(b.flat_grad, out=acc)
        # (one scalar read-back per update.  The sync-free form -- clamp on the device and multiply every bucket by the coefficient,
        #  as torch does -- was measured SLOWER here: 290.8 vs 273.7 ms per video step; this step is bound by streaming weights.)
        coef = min(1.0, max_norm / (float(acc.item()) ** 0.5 + 1e-6))
        if coef < 1.0:
            for b in buckets:
                b.flat_grad.mul_(coef)

    def _video(self, key, dev):
        feats, target = self._video_on_device(key, dev, want_target=True)
        return feats.unsqueeze(1), target.view(-1, 1, 1)

    def pretrain(self, fold):
        """VAE alone before the adversarial game (sumgan.py:323-360)."""
        train_keys, _ = self._get_train_test_keys(fold)
        dev = self._device()
        vae = self.model.summarizer.vae
        opt = FlatAdam(vae.parameters(), lr=self.hps.lr, weight_decay=self.hps.weight_decay)
        for epoch in range(self.pretrain_vae):
            losses = []
            random.shuffle(train_keys)
            for key in train_keys:
                x, _ = self._video(key, dev)
                x_hat, (mu, logvar) = vae(x)
                loss = self.loss_vae(x, x_hat, mu, logvar)
                opt.zero_grad()
                loss.backward()
                self._clip_all([opt])
                opt.step()
                losses.append(loss.detach())
            if epoch % 10 == 0 or epoch == self.pretrain_vae - 1:
                self.log.info(f"Pretrain: {epoch+1:3}/{self.pretrain_vae:3}   Lvae: {float(torch.stack(losses).mean()):.05f}")

    def setup_optimizers(self):
        """The three Adam groups of sumgan.py:372-386 as flat buckets; together they cover every parameter."""
        summ, gan = self.model.summarizer, self.model.gan
        mk = lambda params: FlatAdam(params, lr=self.hps.lr, weight_decay=self.hps.weight_decay)
        self.s_e_optimizer = mk(list(summ.s_lstm.parameters()) + list(summ.vae.e_lstm.parameters()))
        self.d_optimizer = mk(summ.vae.d_lstm.parameters())
        self.c_optimizer = mk(gan.c_lstm.parameters())
        # the buckets inherit whatever gradients the parameters carry: after VAE pre-training the decoder's last (clipped)
        # gradient is still there and enters the first global clip norm, exactly as in the reference (sumgan.py:435)
        self._buckets = [self.s_e_optimizer, self.d_optimizer, self.c_optimizer]

    def _update(self, opt, loss):
        opt.zero_grad()
        loss.backward()
        self._clip_all(self._buckets)
        opt.step()

    # ---- one video, the independent passes of each update batched along the batch axis
    def _generate(self, x, scores_list, eps_list):
        """VAE over several score-weighted copies of one video at once: x (T,1,D), scores_list [(T,1,1)], eps_list
        [(L,1,H)] -> ([x_hat (T,1,D)], h_mu, h_logvar (L,B,H)).  Same arithmetic per copy as `Summarizer.forward`."""
        vae = self.model.summarizer.vae
        xw = torch.cat([x * sc for sc in scores_list], dim=1)
        (h_mu, h_logvar), c = vae.e_lstm(xw)
        h = h_mu + torch.cat(eps_list, dim=1) * torch.exp(0.5 * h_logvar)
        x_hat = vae.d_lstm(x.size(0), h, c)
        return [x_hat[:, i:i + 1] for i in range(len(scores_list))], h_mu, h_logvar

    def _discriminate(self, xs):
        """cLSTM over several sequences at once -> ([probs (1,1)], [h_last (1,H)])."""
        probs, h_last = self.model.gan(torch.cat(xs, dim=1))
        return [probs[i:i + 1] for i in range(len(xs))], [h_last[i:i + 1] for i in range(len(xs))]

    def train_video(self, x, y, noisy):
        """The three updates of one video (sumgan.py:413-479).  x (T,1,D), y (T,1,1) normalised gtscore; `noisy`: multiply the
        discriminator's inputs by Gaussian noise (epoch < epoch_noise).  Returns (Lse, Ld, Lc, D(x), D(x_hat), D(x_hat_p),
        scores) as device tensors.
        The reference runs 5 generator and 8 discriminator passes one after the other; the passes of one update share their
        weights, so they are batched here (3 + 3 passes: every recurrence step then streams the weights once for 2-3
        sequences).  The random draws keep the reference's order -- a draw depends only on its shape, so the reparameterisation
        noise of a pass can be drawn before that pass is computed."""
        summ = self.model.summarizer
        L, H = summ.vae.e_lstm.lstm.num_layers, summ.vae.e_lstm.lstm.hidden_size
        eps_like = torch.empty(L, 1, H, device=x.device)
        T = x.size(0)
        # -- selector and encoder
        scores = summ.s_lstm(x)
        (x_hat,), mu, logvar = self._generate(x, [scores], [torch.randn_like(eps_like)])
        _, (h_real, h_fake) = self._discriminate([x, x_hat])
        sparsity = torch.nn.functional.binary_cross_entropy(scores, y) if self.sup else torch.abs(torch.mean(scores) - self.sigma)
        loss_s_e = self.loss_recons(h_real, h_fake) + self.loss_prior(mu, logvar) + sparsity
        self._update(self.s_e_optimizer, loss_s_e)
        # -- decoder
        scores = summ.s_lstm(x)
        eps1 = torch.randn_like(eps_like); uniform = torch.rand((T, 1, 1)).to(x.device); eps2 = torch.randn_like(eps_like)
        (x_hat, x_hat_p), _, _ = self._generate(x, [scores, uniform], [eps1, eps2])
        (_, probs_fake, probs_uniform), (h_real, h_fake, _) = self._discriminate([x, x_hat, x_hat_p])
        loss_d = self.loss_recons(h_real, h_fake) + self._bce(probs_fake, 0.9) + self._bce(probs_uniform, 0.9)
        self._update(self.d_optimizer, loss_d)
        # -- discriminator
        scores = summ.s_lstm(x)
        eps1 = torch.randn_like(eps_like); uniform = torch.rand((T, 1, 1)).to(x.device); eps2 = torch.randn_like(eps_like)
        (x_hat, x_hat_p), _, _ = self._generate(x, [scores, uniform], [eps1, eps2])
        x_real = x
        if noisy:
            x_real = torch.randn_like(x) * x
            x_hat = x_hat * torch.randn_like(x_hat)
            x_hat_p = x_hat_p * torch.randn_like(x_hat_p)
        (probs_real, probs_fake, probs_uniform), _ = self._discriminate([x_real, x_hat, x_hat_p])
        loss_c = self._bce(probs_real, 0.9) + self._bce(probs_fake, 0.1) + self._bce(probs_uniform, 0.1)
        self._update(self.c_optimizer, loss_c)
        return (loss_s_e.detach(), loss_d.detach(), loss_c.detach(), probs_real.mean().detach(), probs_fake.mean().detach(),
                probs_uniform.mean().detach(), scores.detach())

    def train_video_sequential(self, x, y, noisy):
        """The three updates of one video (sumgan.py:413-479), pass by pass exactly as the reference writes them (13 separate
        generator / discriminator passes).  `train_video` is the batched equivalent used by `train`.  x (T,1,D), y (T,1,1) normalised gtscore; `noisy`: multiply the
        discriminator's inputs by Gaussian noise (epoch < epoch_noise).  Returns (Lse, Ld, Lc, D(x), D(x_hat), D(x_hat_p),
        scores) as device tensors."""
        summ, gan = self.model.summarizer, self.model.gan
        # -- selector and encoder
        x_hat, (mu, logvar), scores = summ(x)
        _, h_real = gan(x)
        _, h_fake = gan(x_hat)
        sparsity = torch.nn.functional.binary_cross_entropy(scores, y) if self.sup else torch.abs(torch.mean(scores) - self.sigma)
        loss_s_e = self.loss_recons(h_real, h_fake) + self.loss_prior(mu, logvar) + sparsity
        self._update(self.s_e_optimizer, loss_s_e)
        # -- decoder
        x_hat, _, _ = summ(x)
        x_hat_p, _, _ = summ(x, uniform=True)
        _, h_real = gan(x)
        probs_fake, h_fake = gan(x_hat)
        probs_uniform, _ = gan(x_hat_p)
        loss_d = self.loss_recons(h_real, h_fake) + self._bce(probs_fake, 0.9) + self._bce(probs_uniform, 0.9)
        self._update(self.d_optimizer, loss_d)
        # -- discriminator
        x_hat, _, scores = summ(x)
        x_hat_p, _, _ = summ(x, uniform=True)
        x_real = x
        if noisy:
            x_real = torch.randn_like(x) * x
            x_hat = x_hat * torch.randn_like(x_hat)
            x_hat_p = x_hat_p * torch.randn_like(x_hat_p)
        probs_real, _ = gan(x_real)
        probs_fake, _ = gan(x_hat)
        probs_uniform, _ = gan(x_hat_p)
        loss_c = self._bce(probs_real, 0.9) + self._bce(probs_fake, 0.1) + self._bce(probs_uniform, 0.1)
        self._update(self.c_optimizer, loss_c)
        return (loss_s_e.detach(), loss_d.detach(), loss_c.detach(), probs_real.mean().detach(), probs_fake.mean().detach(),
                probs_uniform.mean().detach(), scores.detach())

    def train(self, fold):
        self.model.train()
        train_keys, _ = self._get_train_test_keys(fold)
        self.draw_gtscores(fold, train_keys)
        if self.pretrain_vae > 0:
            self.pretrain(fold)
        dev = self._device()
        self.setup_optimizers()
        best = self._fold_best()
        tags = ("Lse", "Ld", "Lc", "D_x", "D_x_hat", "D_x_hat_p")
        for epoch in range(self.hps.epochs):
            log = {t: [] for t in tags}
            dist_scores = {}
            random.shuffle(train_keys)
            for key in train_keys:
                x, y = self._video(key, dev)
                *vals, scores = self.train_video(x, y, noisy=epoch < self.epoch_noise)
                for t, v in zip(tags, vals):
                    log[t].append(v)
                dist_scores[key] = scores
            means = {t: float(torch.stack(v).mean()) for t, v in log.items()}
            kernels.health_check()               # the epoch's host sync: did any persistent recurrence kernel time out?
            self.log.info(f"Epoch: {f'{epoch+1}/{self.hps.epochs}':6}   " + "  ".join(
                f"{n}: {means[t]:.05f}" for n, t in (("Lse", "Lse"), ("Ld", "Ld"), ("Lc", "Lc"), ("D(x)", "D_x"),
                                                     ("D(x_hat)", "D_x_hat"), ("D(x_hat_p)", "D_x_hat_p"))))
            for t in tags:
                self.hps.writer.add_scalar(f"{self.dataset_name}/Fold_{fold+1}/Train/{t}", means[t], epoch)
            self._evaluate_epoch(fold, epoch, best)
        self.draw_scores(fold, dist_scores)
        return best[0], best[1], best[2]
