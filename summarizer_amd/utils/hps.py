"""Minimal hyper-parameter object carrying exactly the fields the Trainer protocol consumes
(summarizer/utils/config.py:23-56,130-146; uses at summarizer/models/__init__.py:12-16,22,29,112,130).
The reference's own `HParameters` works unchanged with these trainers; this helper exists so
tests / bench / standalone scripts need neither tensorboard nor the reference checkout."""
import logging
import types


class NullWriter:
    """SummaryWriter stand-in: records scalars, drops histograms."""
    def __init__(self):
        self.scalars = {}

    def add_scalar(self, tag, value, step=None):
        self.scalars.setdefault(tag, []).append((step, float(value)))

    def add_histogram(self, *a, **k):
        pass

    def add_hparams(self, *a, **k):
        pass

    def close(self):
        pass


def make_hps(dataset, splits, splits_file="splits/synthetic_splits.json", dataset_name="synthetic", **over):
    log = logging.getLogger("summarizer_amd")
    if not log.handlers:
        log.addHandler(logging.NullHandler())
    hps = types.SimpleNamespace(
        use_cuda=True, cuda_device=0, weight_decay=0.00001, lr=0.00005, epochs=10, test_every_epochs=2,
        summary_proportion=0.15, selection_algorithm="knapsack", extra_params={}, logger=log, writer=NullWriter(),
        splits_files=[splits_file], dataset_of_file={splits_file: dataset}, dataset_name_of_file={splits_file: dataset_name},
        splits_of_file={splits_file: splits})
    for k, v in over.items():
        setattr(hps, k, v)
    return hps
