"""hipvae -- host side of the MI355X-native ConvVAE hot path.

PyTorch is used for device memory, streams and torch.distributed only; all
arithmetic runs in libvaenpvc_hip.so (hand-written gfx950 HIP kernels) through the
C-ABI of include/vaenpvc.h.  There is NO CPU fallback: importing `hipvae.lib`
without the built library raises.
"""
from .lib import load_library, HipVaeError  # noqa: F401
from .engine import Engine, arch_to_struct, glorot_init  # noqa: F401
