"""Mirror of lib/layer_utils/projection.py:124-136 (`Projection`)."""
from .. import ops


class Projection(object):
    """`Projection.apply(label, lin_indices_3d, lin_indices_2d, volume_dims)` -> (C,Z,Y,X) fp32.

    label: (C,h,w) or (h,w) feature map; the index tensors are the packed int64 lists of
    ProjectionHelper.compute_projection (slot 0 = count).  Forward only."""

    @staticmethod
    def apply(label, lin_indices_3d, lin_indices_2d, volume_dims):
        return ops.projection(label, lin_indices_3d, lin_indices_2d, volume_dims)

    forward = apply
