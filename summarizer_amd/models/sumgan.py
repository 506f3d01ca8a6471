"""SumGAN's LSTM modules on MI355X (`summarizer/models/sumgan.py`): the sLSTM selector -- the scorer, `SumGAN.forward` is
exactly `s_lstm(x)` (sumgan.py:23-46,251-258) -- and the forward-running stacks of the VAE / GAN side: eLSTM (sumgan.py:48-73)
and cLSTM / GAN (sumgan.py:213-250).  Same constructors, forward signatures and state_dict keys; every module is
differentiable through the HIP backward kernels, so a reference-style training loop can call `.backward()` on losses built
from their outputs.  The GAN training harness itself (SumGANTrainer, sumgan.py:262-533) is not mirrored."""
import torch
import torch.nn as nn

from .. import kernels
from ._bilstm import pack_time_major, bilstm_scores, lstm_stack


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
        h_last = h_n[-1]                                   # output[-1] of the top layer (sumgan.py:231)
        probs = FrameHeadFunction.apply(h_last, self.out[0].weight, self.out[0].bias)
        return probs.view(-1, 1), h_last


class GAN(nn.Module):
    def __init__(self, input_size=1024, hidden_size=1024, num_layers=2):
        """GAN: discriminator."""
        super().__init__()
        self.c_lstm = cLSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)

    def forward(self, x):
        return self.c_lstm(x)
