"""Transformer-encoder scorer on MI355X -- drop-in for the model class of `summarizer/models/transformer.py` (reference),
INFERENCE path (the first "next" model after the north_star scorers).

Same constructor (transformer.py:19), same parameter names / state_dict keys (the stock `nn.TransformerEncoder` objects are
kept as PARAMETER CONTAINERS and never called), same forward contract x (seq_len, batch, input_size) -> (seq_len, batch, 1).
Quirks reproduced: one `layer_norm` used as the encoder's final norm AND after k1 (transformer.py:47,50,100); the in-place
positional add with the sinusoid table's batch re-view (transformer.py:83-89).  Training (`loss.backward()`) is not
implemented for this scorer: calling it with grad enabled raises.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.init as init

from .. import kernels
from .._lib import SumkError
from . import Trainer
from .vasnet import _sinusoid_table


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
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise SumkError("summarizer_amd.Transformer is an inference scorer: wrap the call in torch.no_grad() "
                            "(its backward kernels are not implemented; see DESIGN.md section 7)")
        p = dict(self.named_parameters())
        return kernels.transformer_forward_packed(xp, sb, p, self.encoder_layers, self.attention_heads, self.input_size,
                                                  1e-5, self.epsilon, self.more_residuals, table, rows)


class TransformerTrainer(Trainer):
    """Scoring / evaluation half of the reference trainer (transformer.py:106-124 `_init_model`; `Trainer.test`,
    `predict_dataset`, `load_weights` from the base class).  `train()` is not provided on the HIP path."""
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
        if self.hps.use_cuda:
            torch.cuda.set_device(self.hps.cuda_device)
            model.cuda()
        return model

    def train(self, fold):
        raise SumkError("TransformerTrainer.train: training of the Transformer scorer is not implemented on the HIP path "
                        "(inference/evaluation only); train it with the reference and load the checkpoint with load_weights()")
