"""Drop the HIP forward path into the REFERENCE tree (Sekunde/3D-SIS) without editing it.

    import sys; sys.path.insert(0, "<3D-SIS checkout>"); sys.path.insert(0, "<this repo>/3d-sis_amd")
    import sis3d.dropin; sis3d.dropin.install()
    # ... then exactly what main.py does: cfg_from_file(...); trainval.benchmark(args)

What install() does:
  1. provides the two cffi extension packages the reference imports
     (lib.layer_utils.nms._ext.nms.gpu_nms, lib.layer_utils.roi_pooling._ext.roi_pooling.roi_pooling_forward_cuda)
     with the reference's OWN C signatures (nms_cuda.h, roi_pooling_cuda.h), backed by libsis3d_hip.so;
  2. replaces the legacy instance-style `RoIPoolFunction` (it cannot even be called on torch >= 1.3), `nms`,
     `Projection` with the mirrors in sis3d.layer_utils;
  3. replaces lib.nets.backbones.{ScanNet_Backbone,SUNCG_Backbone,MaskBackbone} with the HIP networks bound to
     the reference's live `cfg`, so `getattr(backbones, cfg.NET)()` / `.init_modules()` / `.load_state_dict()` /
     `.forward(blobs,'TEST',killing_inds)` in lib/model/trainval.py run unchanged.
Pure-Python shims for modules the reference imports but that are not installable offline (easydict, ipdb, ...)
are NOT provided here: they are the caller's environment, not part of the hot path.
"""
import sys
import types

import torch

from . import ops
from .layer_utils import nms_wrapper as _nms_wrapper
from .layer_utils import projection as _projection
from .layer_utils.roi_pooling import roi_pool as _roi_pool


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ---- the reference's cffi entry points, same argument lists ---------------------------------------------------
def gpu_nms(keep, num_out, boxes, nms_overlap_thresh):
    """int gpu_nms(THLongTensor* keep, THLongTensor* num_out, THCudaTensor* boxes, float thresh)
    (lib/layer_utils/nms/src/nms_cuda.h).  keep/num_out are CALLER-ALLOCATED CPU LongTensors, boxes is on the GPU."""
    k, n = ops.nms_raw(boxes, float(nms_overlap_thresh))
    cnt = int(n.item())
    keep[:cnt] = k[:cnt].cpu()
    num_out[0] = cnt
    return 1


def roi_pooling_forward_cuda(pooled_width, pooled_height, pooled_length, spatial_scale, features, rois, output, argmax):
    """int roi_pooling_forward_cuda(int,int,int,float, THCudaTensor* features, rois, output, THCudaIntTensor* argmax)
    (lib/layer_utils/roi_pooling/src/roi_pooling_cuda.h): caller-allocated output/argmax are filled in place."""
    out, arg = ops.roi_pool(features, rois, (int(pooled_width), int(pooled_height), int(pooled_length)), float(spatial_scale))
    output.copy_(out)
    argmax.copy_(arg)
    return 1


def roi_pooling_backward_cuda(pooled_width, pooled_height, pooled_length, spatial_scale, top_grad, rois, bottom_grad, argmax):
    """int roi_pooling_backward_cuda(int,int,int,float, THCudaTensor* top_grad, rois, bottom_grad, THCudaIntTensor* argmax)
    (roi_pooling_cuda.h): OVERWRITES the caller's bottom_grad, as ROIPoolBackward does (`bottom_diff[index] = gradient`,
    roi_pooling_kernel.cu:137-248) -- a reused, non-zeroed buffer gives the reference's result.  The sum over the RoIs that share
    a voxel is built in the reference's order (RoI, then bin, ascending; r6: no atomics): the reference's result bit for bit."""
    g = ops.roi_pool_backward(top_grad, argmax, bottom_grad.shape, channels_last=ops.is_cl(bottom_grad))
    bottom_grad.copy_(g)
    return 1


def install_extension_stubs():
    """Make `from ._ext import nms` / `from ._ext import roi_pooling` of the reference resolve to the HIP library
    (replaces the prebuilt cpython-36 / sm_61 cffi objects under lib/layer_utils/*/_ext)."""
    nms_ns = types.SimpleNamespace(gpu_nms=gpu_nms)
    roi_ns = types.SimpleNamespace(roi_pooling_forward_cuda=roi_pooling_forward_cuda, roi_pooling_backward_cuda=roi_pooling_backward_cuda)
    _mod("lib.layer_utils.nms._ext", nms=nms_ns).__path__ = []
    _mod("lib.layer_utils.nms._ext.nms", gpu_nms=gpu_nms)
    _mod("lib.layer_utils.roi_pooling._ext", roi_pooling=roi_ns).__path__ = []
    _mod("lib.layer_utils.roi_pooling._ext.roi_pooling", roi_pooling_forward_cuda=roi_pooling_forward_cuda,
         roi_pooling_backward_cuda=roi_pooling_backward_cuda)


def install(ref_cfg=None):
    """Patch the (already importable) reference package `lib`.  Returns an undo list for uninstall()."""
    install_extension_stubs()
    import lib.utils.config as rconfig
    cfg = ref_cfg if ref_cfg is not None else rconfig.cfg
    import lib.layer_utils.nms_wrapper as r_nmsw
    import lib.layer_utils.roi_pooling.roi_pool as r_roi
    import lib.layer_utils.projection as r_proj
    import lib.layer_utils.proposal_layer as r_prop
    import lib.nets.network as r_net
    import lib.nets.backbones as r_bb
    from .nets import backbones as hb

    undo = []

    def patch(mod, name, value):
        undo.append((mod, name, getattr(mod, name)))
        setattr(mod, name, value)

    patch(r_nmsw, "nms", _nms_wrapper.nms)
    patch(r_prop, "nms", _nms_wrapper.nms)
    patch(r_roi, "RoIPoolFunction", _roi_pool.RoIPoolFunction)
    patch(r_net, "RoIPoolFunction", _roi_pool.RoIPoolFunction)
    patch(r_proj, "Projection", _projection.Projection)
    patch(r_net, "Projection", _projection.Projection)
    patch(r_proj, "ProjectionHelper", _projection.ProjectionHelper)
    r_tv = sys.modules.get("lib.model.trainval")          # `from ... import ProjectionHelper` binds by name (trainval.py:17)
    if r_tv is not None and hasattr(r_tv, "ProjectionHelper"):
        patch(r_tv, "ProjectionHelper", _projection.ProjectionHelper)

    def bind(cls):
        class _Bound(cls):
            def __init__(self, *a, **k):
                k.setdefault("cfg", cfg)
                super().__init__(*a, **k)
        _Bound.__name__ = cls.__name__
        _Bound.__qualname__ = cls.__name__
        return _Bound

    for name in ("ScanNet_Backbone", "SUNCG_Backbone", "MaskBackbone"):
        patch(r_bb, name, bind(getattr(hb, name)))
    return undo


def uninstall(undo):
    """restore every symbol install() replaced (tests)"""
    for mod, name, old in reversed(undo):
        setattr(mod, name, old)
