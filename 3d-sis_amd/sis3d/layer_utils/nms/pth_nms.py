"""Mirror of lib/layer_utils/nms/pth_nms.py:48-64 (`pth_nms`); `cpu_nms` is deliberately absent."""
from ... import ops


def pth_nms(dets, thresh):
    return ops.nms(dets, thresh)
