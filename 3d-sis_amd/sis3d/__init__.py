"""sis3d -- MI355X-native forward path of 3D-SIS (hand-written HIP for gfx950).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all hot-path arithmetic runs in libsis3d_hip.so (C ABI: include/sis3d.h).  There
is NO CPU fallback: any op called without the library, or on CPU tensors, raises.
"""
import os as _os
import sys as _sys

__all__ = ["config", "synthetic", "ops", "layer_utils", "nets", "parallel"]


def _ask_for_hw_queues():
    """Four chunk pipelines per GPU (engine.PipelinedEngines, scene.SceneRunner) need more hardware queues than HIP's default of 4:
    each pipeline's stream, the capture / copy stream and the null stream must not share one (measured: 1.80 -> 2.28 G voxels/s,
    profiles/r04_hw_queues.txt).  The HIP runtime reads GPU_MAX_HW_QUEUES once, when it initialises (the first HIP call of the
    process), so the package asks for 8 at import time -- only if the variable is unset and torch has not initialised HIP yet;
    an integrator who imports torch.cuda first exports it in the launcher (engine.check_hw_queues warns when it is too low)."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return False
    t = _sys.modules.get("torch")
    if t is not None:
        try:
            if t.cuda.is_initialized():
                return False
        except Exception:
            return False
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    return True


HW_QUEUES_SET_BY_IMPORT = _ask_for_hw_queues()
