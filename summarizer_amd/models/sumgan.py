"""sLSTM selector (2-layer bidirectional LSTM + per-frame head) on MI355X -- the scorer of the reference's
SumGAN (`summarizer/models/sumgan.py:23-46`; `SumGAN.forward` is exactly `s_lstm(x)`, sumgan.py:251-258).
Same constructor and state_dict keys (`lstm.*`, `out.weight`, `out.bias`).  The VAE/GAN training harness of
sumgan.py:48-533 is out of scope."""
import torch.nn as nn

from .. import kernels
from ._bilstm import pack_time_major, bilstm_scores


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
