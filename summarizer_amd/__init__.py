"""summarizer_amd -- MI355X-native (gfx950) frame-importance scoring engine.

Host-side mirror of the reference's `summarizer/models/` scorer interface (VASNet, DSN, sLSTM and their
Trainers) over hand-written HIP kernels reached through the C ABI in include/sumk.h (libsumk.so, ctypes).
PyTorch is used for device memory, streams, autograd plumbing and torch.distributed only.
"""
__version__ = "0.1.0"

# Reference dotted path -> module of this package that replaces it.  Only the hot-path modules are
# aliased: `summarizer.models.rand / logistic / sumgan_att`, `summarizer.utils.config / io` and `main.py` stay the reference's.
REFERENCE_ALIASES = {
    "summarizer.models.vasnet": "summarizer_amd.models.vasnet",            # VASNet, VASNetTrainer
    "summarizer.models.dsn": "summarizer_amd.models.dsn",                  # DSN, DSNTrainer
    "summarizer.models.sumgan": "summarizer_amd.models.sumgan",            # sLSTM ... SumGAN, SumGANTrainer
    "summarizer.models.transformer": "summarizer_amd.models.transformer",  # Transformer, TransformerTrainer
    "summarizer.utils.eval": "summarizer_amd.utils.eval",                  # upsample, generate_summary, evaluate_*
    "summarizer.utils.knapsack": "summarizer_amd.utils.knapsack",          # knapsack_ortools (native DP, no OR-tools)
}


def install_as_reference():
    """Make the reference's OWN files run on the HIP path without editing them: after this call
    `from summarizer.models.vasnet import VASNetTrainer` (summarizer/utils/config.py:12-18) -- and the same for dsn, sumgan,
    transformer, utils.eval, utils.knapsack -- resolve to this package, so `summarizer/utils/config.py` (HParameters, the
    `-m vasnet|dsn|sumgan|transformer` registry, config.py:68-77), `summarizer/main.py` and `benchmark.py` work unchanged.
    Call it before anything imports `summarizer.utils.config`; idempotent.  Returns the list of installed aliases."""
    import importlib
    import sys
    done = []
    for ref_name, own_name in REFERENCE_ALIASES.items():
        mod = importlib.import_module(own_name)
        have = sys.modules.get(ref_name)
        if have is not None and have is not mod:
            raise ImportError(f"{ref_name} is already imported from {getattr(have, '__file__', '?')}: call "
                              "summarizer_amd.install_as_reference() before importing summarizer.utils.config / summarizer.main")
        sys.modules[ref_name] = mod
        done.append(ref_name)
    return done
