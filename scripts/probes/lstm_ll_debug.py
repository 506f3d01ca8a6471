"""h of one BiLSTM layer with the flag-in-data hand-off (SUMK_LSTM_LL=1) against the counter hand-off (=0): run twice, compare the saved arrays."""
import sys, numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
from summarizer_amd import kernels
from summarizer_amd.models.dsn import DSN
torch.manual_seed(0)
In, H, lens = int(sys.argv[2]), int(sys.argv[3]), [int(v) for v in sys.argv[4].split(",")]
m = DSN(In, H, 1).eval().to("cuda:0")
x = torch.randn(sum(lens), In, device="cuda:0")
sb = kernels.SeqBatch.get(lens, torch.device("cuda:0"))
h, _ = kernels.bilstm_layer_forward(x, sb, dict(m.named_parameters()), "rnn.", 0, H)
np.save(sys.argv[1], h.detach().cpu().numpy())
