"""summarizer_amd -- MI355X-native (gfx950) frame-importance scoring engine.

Host-side mirror of the reference's `summarizer/models/` scorer interface (VASNet, DSN, sLSTM and their
Trainers) over hand-written HIP kernels reached through the C ABI in include/sumk.h (libsumk.so, ctypes).
PyTorch is used for device memory, streams, autograd plumbing and torch.distributed only.
"""
__version__ = "0.1.0"
