"""CPU-side host logic that needs no GPU: precision selectors, ingest / model guards, trainer hyper-parameter parsing,
state_dict key compatibility of the SumGAN containers, the deterministic draw helper of the end-to-end goldens."""
import pytest
import torch

import recipes as R


def test_precision_selector():
    from summarizer_amd import kernels
    from summarizer_amd._lib import SumkError
    assert kernels.precision_code(None) == 0 and kernels.precision_code("fp32") == 0 and kernels.precision_code("bf16x3") == 1
    assert kernels.precision_code("bf16x6") == 2 and kernels.precision_code("bf16") == 3      # plain bf16: the training mode
    with pytest.raises(SumkError):
        kernels.precision_code("fp16")


def test_models_refuse_cpu_tensors():
    from summarizer_amd._lib import SumkError
    from summarizer_amd.ingest import StreamingScorer
    from summarizer_amd.models.dsn import DSN
    from summarizer_amd.models.sumgan import eLSTM, cLSTM, dLSTM
    x = torch.zeros(5, 1, 64)
    for m in (DSN(64, 16, 1), eLSTM(64, 16, 2), cLSTM(64, 16, 2)):
        with pytest.raises(SumkError):
            m(x)
    with pytest.raises(SumkError):
        dLSTM(64, 16, 2)(5, torch.zeros(2, 1, 16), torch.zeros(2, 1, 16))
    with pytest.raises(SumkError):
        StreamingScorer(DSN(64, 16, 1))          # model not on a GPU


def test_sumgan_trainer_parses_extra_params_like_the_reference():
    from summarizer_amd.models.sumgan import SumGANTrainer
    from summarizer_amd.utils.datasets import DictDataset
    from summarizer_amd.utils.hps import make_hps
    ep = {"sigma": "0.25", "input_size": "64", "sLSTM_hidden_size": "16", "edLSTM_hidden_size": "24", "cLSTM_hidden_size": "8",
          "edLSTM_num_layers": "1", "pretrain_vae": "3", "sup": True}
    hps = make_hps(DictDataset({}), [{"train_keys": [], "test_keys": []}], epochs=10, use_cuda=False, extra_params=ep)
    tr = SumGANTrainer(hps, hps.splits_files[0])
    m = tr._init_model()
    assert (tr.sigma, tr.sup, tr.pretrain_vae, tr.epoch_noise) == (0.25, True, 3, 2)      # epoch_noise defaults to int(0.2 * epochs)
    sd = m.state_dict()
    assert sd["summarizer.s_lstm.lstm.weight_hh_l1_reverse"].shape == (64, 16)
    assert sd["summarizer.vae.e_lstm.lstm.weight_ih_l0"].shape == (96, 64) and "summarizer.vae.e_lstm.lstm.weight_ih_l1" not in sd
    assert sd["summarizer.vae.d_lstm.lstm.weight_ih_l0"].shape == (96, 24) and sd["summarizer.vae.d_lstm.recons.weight"].shape == (64, 24)
    assert sd["gan.c_lstm.lstm.weight_ih_l1"].shape == (32, 8) and sd["gan.c_lstm.out.0.weight"].shape == (1, 8)


def test_det_random_is_device_independent_and_counter_based():
    a, b = R.DetRandom(5), R.DetRandom(5)
    t = torch.zeros(3, 4)
    x1, u1 = a.randn_like(t), a.rand((2, 3))
    with b.patch():
        x2, u2 = torch.randn_like(t), torch.rand((2, 3))
    assert torch.equal(x1, x2) and torch.equal(u1, u2) and b.n == 2
    assert not torch.equal(x1, a.randn_like(t))                       # the counter advances
    assert torch.randn_like(t).shape == t.shape and not torch.equal(torch.randn_like(t), x1)   # patch removed afterwards
    assert float(u1.min()) >= 0.0 and float(u1.max()) < 1.0


def test_library_binding_imports_torch_first():
    """libsumk.so links the system libamdhip64, PyTorch-ROCm bundles its own: the binding must pull torch in before it dlopens the
    library (loading them the other way round leaves two HIP runtimes in the process -- seen as 'no ROCm-capable device is
    detected' from the first kernel launch on a GPU box when build() ran before smoke() in one process)."""
    import subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from summarizer_amd import _lib\n"
            "assert 'torch' in sys.modules, 'summarizer_amd._lib must import torch before loading libsumk.so'\n"
            "_lib.load(); print('OK')") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """A launcher that started 2 ranks while --gpus says 1 (or the reverse) must not produce a line that claims the wrong N."""
    import os, socket, subprocess, sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SUMK_BENCH_ONE_GPU="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
