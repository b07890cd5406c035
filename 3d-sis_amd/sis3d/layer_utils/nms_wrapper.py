"""Mirror of lib/layer_utils/nms_wrapper.py:7-16."""
from .. import ops


def nms(dets, thresh):
    """dets: (N,6) score-sorted boxes on the GPU; thresh: IoU threshold.
    Returns a LongTensor (K,) of kept indices, ascending, on dets' device -- what
    pth_nms (lib/layer_utils/nms/pth_nms.py:48-64) returns.  CPU tensors are rejected:
    the reference's numpy `cpu_nms` fallback is the test oracle here, not a product path."""
    return ops.nms(dets, thresh)
