"""DSN scorer (bidirectional LSTM + per-frame head) on MI355X -- drop-in for `summarizer/models/dsn.py`.

Same constructor (dsn.py:19), same state_dict keys (`rnn.weight_ih_l0[_reverse]`, ..., `out.0.weight`,
`out.0.bias`), same forward contract x (seq_len, batch, input_size) -> (seq_len, batch, 1) (dsn.py:38-47).
"""
import torch
import torch.nn as nn

from .. import kernels
from .._lib import SumkError
from ._bilstm import pack_time_major, bilstm_scores


class DSN(nn.Module):
    """Deep Summarization Network"""
    def __init__(self, input_size=1024, hidden_size=256, num_layers=1, cell="lstm"):
        super().__init__()
        assert cell in ["lstm", "gru"], "cell must be either 'lstm' or 'gru'"          # dsn.py:21
        if cell != "lstm":
            raise SumkError("summarizer_amd.DSN: only cell='lstm' has a HIP kernel (the reference's optional GRU "
                            "cell, dsn.py:28-33, is not on the scored path: DSNTrainer always builds DSN())")
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.rnn = nn.LSTM(input_size, hidden_size, num_layers=num_layers, bidirectional=True)
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
        return bilstm_scores(self, x_packed, sb, "rnn.", self.num_layers, self.hidden_size, "out.0.weight", "out.0.bias")
