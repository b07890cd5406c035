"""ctypes binding of libsis3d_hip.so (the C-ABI drop-in boundary, include/sis3d.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsis3d_hip.so")

c_int, c_i64, c_f32, c_sz, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes): exactly the declarations of include/sis3d.h
SIGNATURES = {
    "sis3d_abi_version": (c_int, []),
    "sis3d_strerror": (ctypes.c_char_p, [c_int]),
    "sis3d_last_hip_error": (ctypes.c_char_p, []),
    "sis3d_nms_workspace_bytes": (c_sz, [c_int]),
    "sis3d_nms": (c_int, [c_vp, c_int, c_f32, c_int, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "sis3d_scene_merge_workspace_bytes": (c_sz, [c_int, c_int]),
    "sis3d_scene_merge": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_f32, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sis3d_nms_mask": (c_int, [c_vp, c_int, c_f32, c_vp, c_vp]),
    "sis3d_nms_select": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sis3d_roi_pool_forward": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_int,
                                       c_int, c_f32, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "sis3d_roi_pool_levels": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_int,
                                      c_int, c_f32, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "sis3d_roi_pool_backward": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_i64,
                                        c_i64, c_i64, c_i64, c_vp]),
    "sis3d_projection_backward_workspace_bytes": (c_sz, [c_i64]),
    "sis3d_projection_backward": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "sis3d_projection_forward": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "sis3d_project_views_workspace_bytes": (c_sz, [c_int, c_int, c_i64, c_i64]),
    "sis3d_project_views_max": (c_int, [c_vp, c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_i64,
                                        c_i64, c_i64, c_vp, c_sz, c_vp]),
    "sis3d_compute_projection_workspace_bytes": (c_sz, [c_int, c_i64]),
    "sis3d_compute_projection": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_f32,
                                         c_f32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sis3d_upload_f32": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    "sis3d_mail_fetch": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "sis3d_mail_upload": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_vp]),
    "sis3d_mail_post": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "sis3d_tsdf_encode": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_f32, c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp]),
    "sis3d_proposal_decode": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "sis3d_topk_desc": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "sis3d_pack_records": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp]),
    "sis3d_pack_records_post": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp,
                                        c_vp, c_vp]),
    "sis3d_proposal_decode2": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_f32, c_f32, c_f32,
                                       c_vp, c_vp, c_vp, c_vp]),
    "sis3d_softmax2": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "sis3d_classifier_workspace_floats": (c_sz, [c_int, c_int, c_int]),
    "sis3d_classifier_forward": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp,
                                         c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sis3d_classifier_forward_n": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp,
                                           c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sis3d_classifier16_workspace_floats": (c_sz, [c_int, c_int, c_int]),
    "sis3d_classifier16_forward": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp,
                                           c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sis3d_conv_packed_floats": (c_sz, [c_int, c_int, c_int]),
    "sis3d_conv_pack_weight": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "sis3d_conv3d": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp,
                             c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "sis3d_conv3d_chain": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_int,
                                   c_vp, c_vp]),
    "sis3d_conv3d_chain_projected": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp,
                                             c_int, c_int, c_vp, c_vp]),
    "sis3d_conv3d_pw_chain": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_int,
                                      c_int, c_vp, c_vp]),
    "sis3d_conv_pw16_packed_floats": (c_sz, [c_int, c_int]),
    "sis3d_conv_pw16_pack_weight": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "sis3d_conv3d_pw16": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int,
                                  c_int, c_vp, c_int, c_vp]),
    "sis3d_conv3d_k2s2_pw16": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp,
                                       c_int, c_int, c_vp, c_int, c_vp]),
    "sis3d_conv3d_stem_planar2": (c_int, [c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp,
                                          c_int, c_int, c_vp, c_int, c_vp]),
    "sis3d_rpn_heads": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int,
                                c_int, c_vp]),
    "sis3d_conv3d_k3t16_set_trace": (c_int, [c_vp, c_int]),
    "sis3d_conv_k3t16_packed_floats": (c_sz, [c_int, c_int]),
    "sis3d_conv_k3t16_pack_weight": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "sis3d_conv3d_k3wino_prefer": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sis3d_conv_k3wino_packed_floats": (c_sz, [c_int, c_int]),
    "sis3d_conv_k3wino_pack_weight": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "sis3d_conv3d_k3wino": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_vp]),
    "sis3d_conv3d_k3wino_piggyback": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int,
                                              c_vp, c_vp, c_i64, c_vp]),
    "sis3d_ragged_tiling_k3wino": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "sis3d_conv3d_k3wino_ragged": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_i64, c_vp]),
    "sis3d_conv3d_k3wino_ragged_mini": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_i64, c_vp]),
    "sis3d_ragged_tiling_k3wino_mini": (c_int, [c_int, c_int, c_vp, c_vp, c_vp]),
    "sis3d_conv3d_k3t16_brick": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sis3d_conv3d_k3t16": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int,
                                   c_vp]),
    "sis3d_project_views_prepare": (c_int, [c_vp, c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "sis3d_conv3d_batched": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int,
                                     c_vp, c_int, c_int, c_vp]),
    "sis3d_conv3d_planar2": (c_int, [c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "sis3d_ragged_tiling": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "sis3d_conv3d_ragged": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_i64, c_vp]),
    "sis3d_ragged_tiling_k3t16": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "sis3d_conv3d_k3t16_ragged": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_i64, c_int, c_vp]),
    "sis3d_bottleneck16_brick": (c_int, [c_int, c_int, c_int, c_int]),
    "sis3d_bottleneck16": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp,
                                   c_int, c_vp, c_int, c_vp]),
    "sis3d_bottleneck_wino_prefer": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sis3d_bottleneck_wino": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp,
                                      c_int, c_vp, c_vp]),
    "sis3d_conv3d_planar2_ragged": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_i64, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    "sis3d_maxpool3d_3x3x3": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp]),
    "sis3d_conv3d_k2s2_projected_sparse_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "sis3d_conv3d_k2s2_projected_sparse": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp,
                                                   c_int, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "sis3d_enet_initial": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "sis3d_enet_conv1": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "sis3d_enet_block": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                                 c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "sis3d_planar_to_cl": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp]),
    "sis3d_cl_to_planar": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp]),
}

class PwStage(ctypes.Structure):
    """struct sis3d_pw_stage (include/sis3d.h)"""
    _fields_ = [("packed_w", c_vp), ("bias", c_vp), ("residual", c_vp), ("out", c_vp),
                ("cin", c_int), ("cout", c_int), ("res_stride", c_int), ("out_stride", c_int), ("flags", c_int)]


_lib = None


class Sis3dError(RuntimeError):
    pass


def lib():
    """Load the HIP library or fail loudly -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Sis3dError("libsis3d_hip.so is not built (%s). Run `python 3d-sis_amd/build.py` "
                             "or __graft_entry__.build(); sis3d has no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        l = lib()
        raise Sis3dError("%s failed: %s (%d) hip: %s" % (what, l.sis3d_strerror(rc).decode(), rc,
                                                         l.sis3d_last_hip_error().decode()))
