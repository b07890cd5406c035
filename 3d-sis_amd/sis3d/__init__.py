"""sis3d -- MI355X-native forward path of 3D-SIS (hand-written HIP for gfx950).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all hot-path arithmetic runs in libsis3d_hip.so (C ABI: include/sis3d.h).  There
is NO CPU fallback: any op called without the library, or on CPU tensors, raises.
"""
import os as _os
import sys as _sys

__all__ = ["config", "synthetic", "ops", "layer_utils", "nets", "parallel"]


# r6: importing the package does NOT touch os.environ any more (r4-r5 set GPU_MAX_HW_QUEUES=8 here).  The chunk pipelines run on
# streams the engine creates itself and whose hardware queues it verifies (engine.distinct_queue_streams): four pipelines get four
# distinct queues on HIP's default of four; a process that cannot give every pipeline a queue gets engine.check_hw_queues' error /
# warning with the fix spelled out (export GPU_MAX_HW_QUEUES=8 before the process starts).
HW_QUEUES_SET_BY_IMPORT = False
