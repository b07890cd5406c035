"""Mirror of lib/layer_utils/roi_pooling/roi_pool.py:9-38."""
import torch

from ... import ops


class RoIPoolFunction(object):
    """`RoIPoolFunction(pooled_width, pooled_height, pooled_length, spatial_scale)(features, rois)`.

    The reference is a legacy instance-style autograd Function (it raises on torch >= 1.3); the call
    syntax, argument meaning and the attributes it leaves behind (.argmax int32, .rois, .feature_size)
    are kept.  features: (1,C,W,H,L) fp32 on the GPU, NCDHW or channels_last_3d memory; rois: (R,6) in
    scene coordinates.  Returns (R,C,pw,ph,pl).  Unlike the reference -- whose C layer returns 0 on a bad
    shape and Python ignores it (roi_pooling_cuda.c:20-32) -- bad shapes raise.  `backward(grad_output)` is the
    legacy-style manual call of the reference (roi_pool.py:40-50)."""

    def __init__(self, pooled_width, pooled_height, pooled_length, spatial_scale):
        self.pooled_height = int(pooled_height)
        self.pooled_width = int(pooled_width)
        self.pooled_length = int(pooled_length)
        self.spatial_scale = float(spatial_scale)
        self.argmax = None
        self.rois = None
        self.feature_size = None
        self._features_cl = False

    def forward(self, features, rois):
        out, arg = ops.roi_pool(features, rois, (self.pooled_width, self.pooled_height, self.pooled_length),
                                self.spatial_scale, want_argmax=True)
        self.argmax = arg
        self.rois = rois
        self.feature_size = features.size()
        self._features_cl = ops.is_cl(features)
        return out

    __call__ = forward

    def backward(self, grad_output):
        """roi_pool.py:40-50: (grad_input, zeros_like(rois)); grad_input has the memory layout the features had"""
        if self.feature_size is None or self.argmax is None:
            raise ops._lib.Sis3dError("RoIPoolFunction.backward before forward")
        grad_input = ops.roi_pool_backward(grad_output, self.argmax, self.feature_size, channels_last=self._features_cl)
        return grad_input, torch.zeros_like(self.rois)
