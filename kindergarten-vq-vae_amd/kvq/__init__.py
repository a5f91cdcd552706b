"""kvq -- host side of the MI355X-native Kindergarten-VQ-VAE hot path.

Everything numeric on the path runs in libkvq.so (hand-written HIP for gfx950, C ABI in include/kvq.h);
this package is the thin PyTorch-ROCm plumbing around it: device memory, streams, autograd glue,
torch.distributed.  There is no CPU or eager fallback: if the library or a GPU is missing the ops raise.
"""
from . import _ffi  # noqa: F401
from .functional import (vq_forward_backward_available, vector_quantize, fused_cross_entropy,  # noqa: F401
                         vq_one_hot, vq_ema_update, vq_debug_distances)

__all__ = ["vector_quantize", "fused_cross_entropy", "vq_one_hot", "vq_ema_update", "vq_debug_distances",
           "vq_forward_backward_available"]
