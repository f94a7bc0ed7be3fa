"""de-i2i-gan on MI355X: the defectGAN G+D train step of jason2714/de-i2i-gan behind the reference's own
Generator / Discriminator / Model / Trainer surface, computed by hand-written HIP kernels for gfx950
(``csrc/`` -> ``lib/libdei2i_hip.so``, C ABI in ``include/dei2i_hip.h``).

Host code is Python on PyTorch-ROCm; PyTorch supplies device memory, streams, autograd bookkeeping and
``torch.distributed`` (RCCL) only.  There is no CPU fallback: importing the ops without the built library,
or calling them on a non-GPU tensor, raises.
"""
__version__ = "0.1.0"
