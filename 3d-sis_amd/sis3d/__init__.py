"""sis3d -- MI355X-native forward path of 3D-SIS (hand-written HIP for gfx950).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all hot-path arithmetic runs in libsis3d_hip.so (C ABI: include/sis3d.h).  There
is NO CPU fallback: any op called without the library, or on CPU tensors, raises.
"""
__all__ = ["config", "synthetic", "ops", "layer_utils", "nets", "parallel"]
