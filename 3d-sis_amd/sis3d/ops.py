"""torch-tensor front ends of the C ABI (include/sis3d.h).

Every function enqueues on torch's CURRENT stream and never synchronises unless
its return value has a data-dependent shape (documented per function).  Inputs
must be CUDA (ROCm) tensors; CPU tensors raise -- there is no fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib

CL3D = torch.channels_last_3d

EPI_RELU, EPI_RESIDUAL, EPI_SIGMOID, EPI_RPN_HEAD = 1, 2, 4, 8
DISPATCH_SHARED_CHIP = 0x100          # include/sis3d.h SIS3D_DISPATCH_SHARED_CHIP


# ---- dispatch regime (r5): which FORM the k3 convolutions take is a property of the caller -- a launch that has the chip to itself
# takes the form that finishes soonest, launches made while other chunks' kernels are in flight take the form that costs the fewest
# CU-microseconds.  It travels to the library as per-call ARGUMENTS (the flag above, sis3d_*_prefer's
# shared_chip, sis3d_conv3d_k3t16_brick's max_voxels); on this side it is a THREAD-LOCAL value set by `dispatch_regime(...)`, so two
# threads that capture different regimes at the same time do not see each other's setting and nothing has to be reset afterwards.
import contextlib as _contextlib
import threading as _threading

_REGIME = _threading.local()


def regime():
    """(shared_chip: bool, brick_cap: int voxels, 0 = none) of the calling thread"""
    return getattr(_REGIME, "value", (False, 0))


@_contextlib.contextmanager
def dispatch_regime(shared_chip=False, brick_cap=0):
    """launches made by THIS thread inside the block are dispatched for a shared chip (several chunks in flight) / with the k3 direct
    kernel's brick capped at `brick_cap` voxels; the previous regime of the thread is restored on exit (nesting is fine)"""
    old = regime()
    _REGIME.value = (bool(shared_chip), int(brick_cap))
    try:
        yield
    finally:
        _REGIME.value = old


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.Sis3dError("%s must be a CUDA/ROCm tensor (sis3d has no CPU path)" % name)
    if t.dtype != dtype:
        raise _lib.Sis3dError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ---------------------------------------------------------------------- NMS --
def nms_raw(dets, thresh, max_keep=0, path=0):
    """-> (keep int64 [n], num_keep int32 [1]) on the device; no host sync.  path: 0 = algorithm by size, 1 = one-workgroup sweep,
    2 = sparse suppressor table + parallel resolve (same keep list; the parity tests force both)."""
    dets = _dev(dets, "dets").contiguous()
    if dets.dim() != 2 or dets.shape[1] != 6:
        raise _lib.Sis3dError("dets must be (N,6)")
    n = dets.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=dets.device)
    num = torch.empty(1, dtype=torch.int32, device=dets.device)      # written by the sweep kernel on every path
    wsb = lib().sis3d_nms_workspace_bytes(n)
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dets.device)
    check(lib().sis3d_nms(_ptr(dets), n, float(thresh), int(max_keep), _ptr(keep), _ptr(num), _ptr(ws), wsb, int(path), _stream()), "sis3d_nms")
    return keep, num


def nms(dets, thresh, max_keep=0, path=0):
    """lib/layer_utils/nms_wrapper.py:7-16 semantics: LongTensor (K,) of kept indices on dets' device.
    The result length is data dependent -> one 4-byte D2H read (the reference copies the whole
    bit matrix and sweeps on the host)."""
    keep, num = nms_raw(dets, thresh, max_keep, path)
    return keep[:int(num.item())].contiguous()


def scene_merge_raw(blocks, k_rows, thresh, score_col=6, box_col=0, max_keep=0):
    """sis3d_scene_merge on gathered record blocks (n_chunks, 1 + k_rows*W): -> (recs (T,W) padded, order int32 (T,),
    keep int64 (T,), counts int32 [valid rows, kept]) on the device, no host sync."""
    blocks = _dev(blocks, "blocks").contiguous()
    n_chunks, bf = blocks.shape
    width = (bf - 1) // int(k_rows)
    if 1 + width * int(k_rows) != bf:
        raise _lib.Sis3dError("blocks must be (n_chunks, 1 + k_rows*width)")
    T = n_chunks * int(k_rows)
    dev = blocks.device
    recs = torch.empty(T, width, device=dev)
    order = torch.empty(T, dtype=torch.int32, device=dev)
    keep = torch.empty(T, dtype=torch.int64, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    wsb = lib().sis3d_scene_merge_workspace_bytes(n_chunks, int(k_rows))
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    check(lib().sis3d_scene_merge(_ptr(blocks), n_chunks, int(k_rows), width, int(score_col), int(box_col), float(thresh),
                                  int(max_keep), _ptr(recs), _ptr(order), _ptr(keep), _ptr(counts), _ptr(ws), wsb, _stream()),
          "sis3d_scene_merge")
    return recs, order, keep, counts


def scene_merge(blocks, k_rows, thresh, score_col=6, box_col=0, max_keep=0):
    """parallel.merge_scene as one device sequence: -> (records sorted by score (N,W), keep LongTensor, chunk id of every
    sorted record).  One 8-byte readback for the two data-dependent lengths."""
    recs, order, keep, counts = scene_merge_raw(blocks, k_rows, thresh, score_col, box_col, max_keep)
    total, kept = counts.tolist()
    return recs[:total], keep[:kept], (order[:total] // int(k_rows)).long()


def nms_mask(dets, thresh):
    dets = _dev(dets, "dets").contiguous()
    n = dets.shape[0]
    mask = torch.zeros(n, (n + 63) // 64, dtype=torch.int64, device=dets.device)
    check(lib().sis3d_nms_mask(_ptr(dets), n, float(thresh), _ptr(mask), _stream()), "sis3d_nms_mask")
    return mask


def nms_select(boxes_all, level_all, scores_sorted, order, n, thresh, max_keep):
    """Fused top-n -> NMS -> first max_keep survivors (proposal_layer.py:181-197).  Fixed-size outputs
    (rows >= num_keep are zero), no host sync."""
    dev = boxes_all.device
    rois = torch.empty(max_keep, 6, device=dev)
    scores = torch.empty(max_keep, device=dev)
    levels = torch.empty(max_keep, device=dev)
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    num = torch.empty(1, dtype=torch.int32, device=dev)               # written by the sweep kernel on every path
    wsb = lib().sis3d_nms_workspace_bytes(n)
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    check(lib().sis3d_nms_select(_ptr(_dev(boxes_all, "boxes_all")), _ptr(_dev(level_all, "level_all")),
                                 _ptr(_dev(scores_sorted, "scores_sorted")), _ptr(_dev(order, "order", torch.int64)), int(n),
                                 float(thresh), int(max_keep), _ptr(rois), _ptr(scores), _ptr(levels), _ptr(keep), _ptr(num),
                                 _ptr(ws), wsb, _stream()), "sis3d_nms_select")
    return rois, scores, levels, keep, num


# ------------------------------------------------------------------ RoI pool --
def roi_pool(features, rois, pooled, scale, want_argmax=True, out_channels_last=False):
    """features: logical (1,C,W,H,L), any strides (NCDHW or channels_last_3d).  -> out (R,C,pw,ph,pl)
    [+ argmax int32].  out_channels_last: memory order (R, bins, C) behind the same logical shape."""
    f = _dev(features, "features")
    r = _dev(rois, "rois").contiguous()
    if f.dim() != 5 or f.shape[0] != 1:
        raise _lib.Sis3dError("features must be (1,C,W,H,L): batch size 1 only (roi_pooling_cuda.c:29-32)")
    if r.dim() != 2 or r.shape[1] != 6:
        raise _lib.Sis3dError("rois must be (R,6) (roi_pooling_cuda.c:20-23)")
    _, C, W, H, L = f.shape
    R = r.shape[0]
    pw, ph, pl = pooled
    nb = pw * ph * pl
    if out_channels_last:
        out = torch.empty(R, pw, ph, pl, C, device=f.device).permute(0, 4, 1, 2, 3)
        os_n, os_c, os_b = nb * C, 1, C
    else:
        out = torch.empty(R, C, pw, ph, pl, device=f.device)
        os_n, os_c, os_b = C * nb, nb, 1
    arg = None
    if want_argmax:
        arg = torch.empty(out.shape, dtype=torch.int32, device=f.device) if not out_channels_last else \
            torch.empty(R, pw, ph, pl, C, dtype=torch.int32, device=f.device).permute(0, 4, 1, 2, 3)
    st = f.stride()
    check(lib().sis3d_roi_pool_forward(_ptr(f), C, W, H, L, st[1], st[2], st[3], st[4], _ptr(r), R, pw, ph, pl, float(scale),
                                       _ptr(out), _ptr(arg), os_n, os_c, os_b, _stream()), "sis3d_roi_pool_forward")
    return (out, arg) if want_argmax else out


def roi_pool_backward(grad_output, argmax, feature_size, channels_last=False):
    """ROIPoolBackward (roi_pooling_kernel.cu:137-248): scatter grad_output through the argmax saved by roi_pool.
    grad_output / argmax: (R,C,pw,ph,pl), any strides (both the same); -> grad_input, logical feature_size (1,C,W,H,L)."""
    g = _dev(grad_output, "grad_output")
    a = _dev(argmax, "argmax", torch.int32)
    if g.shape != a.shape or g.dim() != 5:
        raise _lib.Sis3dError("grad_output and argmax must both be (R,C,pw,ph,pl)")
    if g.stride() != a.stride():
        g = g.contiguous()
        a = a.contiguous()
    R, C, pw, ph, pl = g.shape
    _, C2, W, H, L = (int(v) for v in feature_size)
    if C2 != C or int(feature_size[0]) != 1:
        raise _lib.Sis3dError("feature_size must be (1,%d,W,H,L)" % C)
    gin = new_act(C, (W, H, L), g.device).zero_() if channels_last else torch.zeros(1, C, W, H, L, device=g.device)
    st, gs = g.stride(), gin.stride()
    if st[3] != pl * st[4] or st[2] != ph * st[3]:
        raise _lib.Sis3dError("roi_pool_backward: bins must be contiguous in (pw,ph,pl) order")
    check(lib().sis3d_roi_pool_backward(_ptr(g), _ptr(a), R, C, pw, ph, pl, st[0], st[1], st[4], W, H, L, _ptr(gin), gs[1], gs[2], gs[3],
                                        gs[4], _stream()), "sis3d_roi_pool_backward")
    return gin


def projection_backward(grad_output, lin3d, lin2d, image_hw=(32, 41)):
    """Projection.backward (projection.py:139-153): (C,Z,Y,X) gradient -> (C,h,w) gradient of the label image, with the
    reference's clone-and-resize initial values (see sis3d_projection_backward)."""
    g = _dev(grad_output, "grad_output").contiguous()
    C = g.shape[0]
    nvox = g[0].numel()
    a = _dev(lin3d, "lin_indices_3d", torch.int64).contiguous()
    b = _dev(lin2d, "lin_indices_2d", torch.int64).contiguous()
    npix = int(image_hw[0]) * int(image_hw[1])
    out = torch.empty(C, int(image_hw[0]), int(image_hw[1]), device=g.device)
    wsb = lib().sis3d_projection_backward_workspace_bytes(npix)
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=g.device)
    check(lib().sis3d_projection_backward(_ptr(g), C, nvox, _ptr(a), _ptr(b), npix, _ptr(out), _ptr(ws), wsb, _stream()),
          "sis3d_projection_backward")
    return out


def roi_pool_levels(f1, f2, rois, levels, pooled, scale, out_channels_last=True):
    f1, f2 = _dev(f1, "features1"), _dev(f2, "features2")
    if f1.shape != f2.shape or f1.stride() != f2.stride():
        raise _lib.Sis3dError("both pyramid levels must share shape and strides")
    r = _dev(rois, "rois").contiguous()
    lv = _dev(levels, "levels").contiguous()
    _, C, W, H, L = f1.shape
    R = r.shape[0]
    nb = pooled ** 3
    if out_channels_last:
        out = torch.empty(R, pooled, pooled, pooled, C, device=f1.device).permute(0, 4, 1, 2, 3)
        os_n, os_c, os_b = nb * C, 1, C
    else:
        out = torch.empty(R, C, pooled, pooled, pooled, device=f1.device)
        os_n, os_c, os_b = C * nb, nb, 1
    st = f1.stride()
    check(lib().sis3d_roi_pool_levels(_ptr(f1), _ptr(f2), C, W, H, L, st[1], st[2], st[3], st[4], _ptr(r), _ptr(lv), R, pooled,
                                      float(scale), _ptr(out), os_n, os_c, os_b, _stream()), "sis3d_roi_pool_levels")
    return out


# ---------------------------------------------------------------- projection --
def projection(label, lin3d, lin2d, volume_dims):
    """Projection.forward (projection.py:124-136): -> (C,Z,Y,X)."""
    label = _dev(label, "label")
    C = 1 if label.dim() == 2 else label.shape[0]
    feat = label.contiguous().view(C, -1)
    a = _dev(lin3d, "lin_indices_3d", torch.int64).contiguous()
    b = _dev(lin2d, "lin_indices_2d", torch.int64).contiguous()
    X, Y, Z = (int(v) for v in volume_dims)
    nvox = X * Y * Z
    if a.numel() < nvox + 1 or b.numel() < nvox + 1:
        raise _lib.Sis3dError("index lists must have volume+1 entries (slot 0 = count)")
    out = torch.empty(C, Z, Y, X, device=label.device)
    check(lib().sis3d_projection_forward(_ptr(feat), C, feat.shape[1], _ptr(a), _ptr(b), nvox, _ptr(out), _stream()),
          "sis3d_projection_forward")
    return out


def project_views_max(feats, lin3d, lin2d, volume_dims, killing_inds=(), channels_last=True):
    """Fused network.py:216-239.  -> logical (1,C,X,Y,Z); memory order channels-last (for our conv stack)
    or the reference's (C,Z,Y,X) (permute(4,0,3,2,1) of a (C,Z,Y,X,1) tensor)."""
    feats = _dev(feats, "feats").contiguous()
    a = _dev(lin3d, "proj_ind_3d", torch.int64).contiguous()
    b = _dev(lin2d, "proj_ind_2d", torch.int64).contiguous()
    X, Y, Z = (int(v) for v in volume_dims)
    nvox = X * Y * Z
    V = min(feats.shape[0], a.shape[0], b.shape[0])      # zip() truncation of the reference loop
    C = feats.shape[1]
    npix = feats.shape[2] * feats.shape[3]
    kill = (ctypes.c_uint8 * V)(*[1 if k in set(killing_inds or ()) else 0 for k in range(V)])
    wsb = lib().sis3d_project_views_workspace_bytes(V, C, npix, nvox)
    ws = torch.empty(wsb, dtype=torch.uint8, device=feats.device)
    if channels_last:
        out = new_act(C, (X, Y, Z), feats.device)
        os_c, os_x, os_y, os_z = 1, Y * Z * C, Z * C, C
    else:
        out = torch.empty(C, Z, Y, X, 1, device=feats.device).permute(4, 0, 3, 2, 1)
        os_c, os_x, os_y, os_z = nvox, 1, X, X * Y
    check(lib().sis3d_project_views_max(_ptr(feats), V, C, npix, _ptr(a), _ptr(b), kill, X, Y, Z, _ptr(out), os_c, os_x, os_y,
                                        os_z, _ptr(ws), wsb, _stream()), "sis3d_project_views_max")
    return out


class ProjectedVolume(object):
    """The back-projected image volume WITHOUT the volume: a voxel->pixel table per included view plus pixel-major
    feature rows (sis3d_project_views_prepare).  Logical shape (1,C,X,Y,Z); `conv3d_chain` reads it directly
    (sis3d_conv3d_chain_projected), `dense()` materialises the tensor `project_views_max` would have produced."""

    def __init__(self, table, rows, nslots, C, npix, dims, src):
        self.table, self.rows, self.nslots, self.C, self.npix, self.dims, self._src = table, rows, nslots, C, npix, tuple(dims), src
        self.shape = (1, C) + self.dims
        self.device = table.device

    def dense(self, channels_last=True):
        feats, a, b, kill = self._src
        return project_views_max(feats, a, b, self.dims, kill, channels_last=channels_last)


def project_views_prepare(feats, lin3d, lin2d, volume_dims, killing_inds=()):
    """network.py:216-239 as (table, feature rows) -> ProjectedVolume; same arguments as project_views_max."""
    feats = _dev(feats, "feats").contiguous()
    a = _dev(lin3d, "proj_ind_3d", torch.int64).contiguous()
    b = _dev(lin2d, "proj_ind_2d", torch.int64).contiguous()
    X, Y, Z = (int(v) for v in volume_dims)
    nvox = X * Y * Z
    V = min(feats.shape[0], a.shape[0], b.shape[0])
    C = feats.shape[1]
    npix = feats.shape[2] * feats.shape[3]
    kills = set(killing_inds or ())
    kill = (ctypes.c_uint8 * V)(*[1 if k in kills else 0 for k in range(V)])
    table = torch.empty(V, nvox, dtype=torch.int32, device=feats.device)
    rows = torch.empty(V, npix, C, device=feats.device)
    ns = ctypes.c_int(0)
    check(lib().sis3d_project_views_prepare(_ptr(feats), V, C, npix, _ptr(a), _ptr(b), kill, nvox, _ptr(table), _ptr(rows),
                                            ctypes.byref(ns), _stream()), "sis3d_project_views_prepare")
    return ProjectedVolume(table, rows, ns.value, C, npix, (X, Y, Z), (feats, a, b, tuple(killing_inds or ())))


VIEW_PARAM_FLOATS = 40


def compute_projection(depths, view_params, volume_dims, image_dims, intrinsic, depth_min, depth_max, voxel_size, out=None):
    """Device form of ProjectionHelper.compute_projection for V views (projection.py:52-121).
    depths (V,H,W) fp32 cuda, view_params (V,40) fp32 cuda (see include/sis3d.h) -> (lin3d, lin2d) int64 (V,nvox+1)."""
    depths = _dev(depths, "depths").contiguous()
    view_params = _dev(view_params, "view_params").contiguous()
    V = depths.shape[0]
    W, H = int(image_dims[0]), int(image_dims[1])
    if depths[0].numel() != W * H or tuple(view_params.shape) != (V, VIEW_PARAM_FLOATS):
        raise _lib.Sis3dError("compute_projection: depths must be (V,%d,%d) and view_params (V,%d)" % (H, W, VIEW_PARAM_FLOATS))
    X, Y, Z = (int(v) for v in volume_dims)
    nvox = X * Y * Z
    if out is None:
        out = (torch.empty(V, nvox + 1, dtype=torch.int64, device=depths.device),
               torch.empty(V, nvox + 1, dtype=torch.int64, device=depths.device))
    wsb = lib().sis3d_compute_projection_workspace_bytes(V, nvox)
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=depths.device)
    check(lib().sis3d_compute_projection(_ptr(depths), _ptr(view_params), V, X, Y, Z, W, H, float(intrinsic[0][0]),
                                         float(intrinsic[1][1]), float(intrinsic[0][2]), float(intrinsic[1][2]), float(depth_min),
                                         float(depth_max), float(voxel_size), _ptr(out[0]), _ptr(out[1]), _ptr(ws), wsb,
                                         _stream()), "sis3d_compute_projection")
    return out


TSDF_MODES = {"abs": 0, "flip": 1, "log": 2}


def _pinned(t, name):
    """a float32 tensor in PINNED host memory (device-mapped: kernels may read it through its host pointer)"""
    if not isinstance(t, torch.Tensor) or t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_pinned():
        raise _lib.Sis3dError("%s must be a contiguous float32 tensor in pinned host memory" % name)
    return t


def upload(host, out, workgroups=0):
    """pinned host tensor -> `out` (cuda, same numel) by a kernel that reads the host memory across PCIe (sis3d_upload_f32): an
    ordinary launch on the current stream -- it never blocks the enqueueing thread, unlike a hipMemcpyAsync behind a pending graph"""
    _pinned(host, "host")
    _dev(out, "out")
    if not out.is_contiguous() or out.numel() != host.numel():
        raise _lib.Sis3dError("upload: `out` must be contiguous with %d elements" % host.numel())
    check(lib().sis3d_upload_f32(_ptr(host), _ptr(out), host.numel(), int(workgroups), _stream()), "sis3d_upload_f32")
    return out


class Mailbox(object):
    """Host side of the host -> graph mailbox (include/sis3d.h, sis3d_mail_upload / sis3d_mail_post): a ring of 64-byte slots in
    PINNED host memory that kernels inside a captured graph read, the device counter of consumed slots and the pinned progress
    word the device writes back.  `write()` is plain CPU stores -- no HIP call -- so a pipeline's only call per chunk is the graph
    launch (a command enqueued behind a graph launch that has not finished can block the host on this runtime).
    slot = { u64 src; u64 dst; f32 origin[3]; u32 flags; u64 next_src; u64 pad[3] }."""
    RING = 256
    SLOT = 64

    def __init__(self, device, ring=RING):
        import numpy as np
        self.ring_size = int(ring)
        self.buf = torch.zeros(self.ring_size * self.SLOT, dtype=torch.uint8).pin_memory()
        a = self.buf.numpy()
        self.u64 = a.view(np.uint64).reshape(self.ring_size, self.SLOT // 8)
        self.f32 = a.view(np.float32).reshape(self.ring_size, self.SLOT // 4)
        self.u32 = a.view(np.uint32).reshape(self.ring_size, self.SLOT // 4)
        # [0] consumed slots, [8..23] the slot of the running pass, [24..25] the source whose chunk the staging buffer holds
        self.state = torch.zeros(32, dtype=torch.int32, device=device)
        self._progress_t = torch.zeros(2, dtype=torch.int64).pin_memory()     # [0] consumed slots, [1] why the device rejected a slot
        self.progress = self._progress_t.numpy()
        self.head = 0                                  # slots written so far (the device has consumed progress[0] of them)
        self.keep = [None] * self.ring_size            # the tensors behind the pointers of the outstanding slots stay alive
        self._released = 0                             # slots below this number have been consumed and released
        self._announced = 0                            # next_src of the slot written last

    def write(self, src=None, dst=None, origin=None, next_src=None):
        """the slot of the NEXT pass: src = tensor to copy into the pipeline's input (pinned host or device memory; None: the input
        buffer already holds the chunk), dst = tensor that receives the pass's record block (None: nowhere), origin = (x, y, z),
        next_src = the PINNED HOST chunk of the pass after this one, if known: the piggyback row of this pass's longest conv launch
        pulls it into the staging buffer (sis3d_conv3d_k3wino_piggyback) -- the caller leaves it unchanged until that pass has run"""
        self.check()
        if self.head - int(self.progress[0]) >= self.ring_size - 1:
            # the producer is a whole ring ahead of the device: wait for a slot (rare); a device that never consumes is an error
            import time
            t_end = time.monotonic() + 60.0
            while self.head - int(self.progress[0]) >= self.ring_size - 1:
                if time.monotonic() > t_end:
                    raise _lib.Sis3dError("mailbox: the device has not consumed a slot for 60 s (%d written, %d consumed)"
                                          % (self.head, int(self.progress[0])))
        k = self.head % self.ring_size
        self.u64[k, 0] = src.data_ptr() if src is not None else 0
        self.u64[k, 1] = dst.data_ptr() if dst is not None else 0
        flags = 2 if (src is not None and src.is_cuda) else 0          # bit 1: a device source is copied by the whole grid
        if src is not None and not src.is_cuda and self._announced == src.data_ptr():
            flags |= 4                                                 # bit 2: the previous pass was told to stage this very chunk
        if origin is not None:
            self.f32[k, 4] = origin[0]; self.f32[k, 5] = origin[1]; self.f32[k, 6] = origin[2]
            flags |= 1
        self._announced = next_src.data_ptr() if (next_src is not None and not next_src.is_cuda) else 0
        self.u64[k, 4] = self._announced
        self.u32[k, 7] = flags
        # integrity stamp (include/sis3d.h): sequence number + check word over the pointers, verified by the fetch kernel
        self.u64[k, 5] = self.head
        self.u64[k, 6] = self.check_word(int(self.u64[k, 0]), int(self.u64[k, 1]), self._announced, flags, self.head)
        self.keep[k] = (src, dst, next_src)
        self.head += 1
        # the tensors behind CONSUMED slots are released (a streaming caller with freshly allocated chunks otherwise keeps
        # ring_size chunks alive per pipeline); a consumed slot's next_src has been pulled or superseded by its successor's src
        done = int(self.progress[0])
        while self._released < done and self._released < self.head - 1:
            self.keep[self._released % self.ring_size] = None
            self._released += 1

    @staticmethod
    def check_word(src, dst, next_src, flags, seq):
        m = (1 << 64) - 1
        rotl = lambda v, r: ((v << r) | (v >> (64 - r))) & m
        return (src ^ rotl(dst, 17) ^ rotl(next_src, 31) ^ ((flags << 40) & m) ^ ((seq * 0x9E3779B97F4A7C15) & m) ^ 0x5151D3D3) & m

    def check(self):
        """raise if the device has rejected a slot (stale / lapped: 1, torn: 2); the pass that fetched it copied nothing"""
        e = int(self.progress[1])
        if e:
            raise _lib.Sis3dError("mailbox: the device rejected a slot (%s); %d written, %d consumed"
                                  % ({1: "stale or lapped sequence number", 2: "check word mismatch"}.get(e, "code %d" % e), self.head,
                                     int(self.progress[0])))


def mail_source(t, numel):
    """a tensor a mailbox slot may name as the source of a chunk: contiguous float32 with `numel` elements, on the device or in pinned
    host memory -> the tensor, else None (the caller copies it the ordinary way)"""
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != numel:
        return None
    if t.is_cuda or t.is_pinned():
        return t
    return None


def mail_upload(mb, input_dst, origin_dst=None, workgroups=0, staged=None):
    """first two nodes of a pipeline's graph: fetch the slot of this pass across PCIe (sis3d_mail_fetch), then slot.src -> input_dst and
    slot.origin -> origin_dst (sis3d_mail_upload).  staged: the piggyback staging buffer (same numel as input_dst) -- copied instead
    of slot.src when the previous pass already pulled this chunk into it"""
    _dev(input_dst, "input_dst")
    if staged is not None and (_dev(staged, "staged").numel() != input_dst.numel()):
        raise _lib.Sis3dError("mail_upload: the staging buffer must have the input's size")
    check(lib().sis3d_mail_fetch(_ptr(mb.buf), mb.ring_size, _ptr(mb.state), _stream()), "sis3d_mail_fetch")
    check(lib().sis3d_mail_upload(_ptr(mb.state), _ptr(input_dst), input_dst.numel(), _ptr(origin_dst), _ptr(staged), int(workgroups),
                                  _stream()), "sis3d_mail_upload")


# piggyback upload (r5): inside `with piggyback(mb, stage)` the first LONG Winograd launch (>= PIGGY_MIN_FLOPS: the rpn_net pair, ~50 us
# alone, ~110 us on a shared chip) carries one more row of workgroups, eight of which pull the mailbox slot's next_src into `stage`
# (sis3d_conv3d_k3wino_piggyback).  Thread-local, like the dispatch regime: engines are prepared concurrently.
PIGGY_MIN_FLOPS = 8e9


@_contextlib.contextmanager
def piggyback(mb, stage):
    """-> a one-element list: [True] once a launch has taken the upload on board"""
    prev = getattr(_REGIME, "piggy", None)
    took = [False]
    _REGIME.piggy = (mb, _dev(stage, "stage"), took) if mb is not None and stage is not None else None
    try:
        yield took
    finally:
        _REGIME.piggy = prev


def mail_post(mb, block_src):
    """last node: the record block -> slot.dst, and the slot is consumed (sis3d_mail_post)"""
    n = block_src.numel() if block_src is not None else 0
    check(lib().sis3d_mail_post(_ptr(mb.state), _ptr(block_src), n, _ptr(mb._progress_t), _stream()), "sis3d_mail_post")


def tsdf_encode(sdf, dims, truncated=3.0, mode="abs", max_height=None, channels_last=True, out=None):
    """raw sdf grid in file order (flat, x fastest; numel X*Y*Z, cuda) -> network input, logical (1,2,X,Yout,Z)
    (dataset.py:54-70 + the max-height crop :196-211).  out: a contiguous planar (1,2,X,Yout,Z) buffer to write into (the static
    input buffer of a chunk pipeline: engine.PipelinedEngines.run_fed)."""
    if isinstance(sdf, torch.Tensor) and not sdf.is_cuda and out is not None:
        sdf = _pinned(sdf, "sdf")                    # r5: the kernel reads the pinned host block itself (upload + encode in one pass)
    else:
        sdf = _dev(sdf, "sdf").contiguous()
    X, Y, Z = (int(v) for v in dims)
    if sdf.numel() != X * Y * Z:
        raise _lib.Sis3dError("tsdf_encode: sdf has %d elements, dims say %d" % (sdf.numel(), X * Y * Z))
    Yo = Y if max_height is None else min(Y, int(max_height))
    if out is not None:
        if tuple(out.shape) != (1, 2, X, Yo, Z) or not out.is_contiguous() or not out.is_cuda or out.dtype != torch.float32:
            raise _lib.Sis3dError("tsdf_encode: `out` must be a contiguous float32 (1,2,%d,%d,%d) cuda tensor" % (X, Yo, Z))
        st = (X * Yo * Z, Yo * Z, Z, 1)
    elif channels_last:
        out = new_act(2, (X, Yo, Z), sdf.device)
        st = (1, Yo * Z * 2, Z * 2, 2)
    else:
        out = torch.empty(1, 2, X, Yo, Z, device=sdf.device)
        st = (X * Yo * Z, Yo * Z, Z, 1)
    check(lib().sis3d_tsdf_encode(_ptr(sdf), X, Y, Z, Yo, float(truncated), TSDF_MODES[mode], _ptr(out), st[0], st[1], st[2],
                                  st[3], _stream()), "sis3d_tsdf_encode")
    return out


# ------------------------------------------------------------------ proposals --
def proposal_decode(anchors, deltas, prob_fg, inside, dims, level_id, out_boxes, out_scores, out_levels):
    n = int(inside.numel())
    check(lib().sis3d_proposal_decode(_ptr(anchors), _ptr(deltas), _ptr(prob_fg), _ptr(inside), n, float(dims[0]), float(dims[1]),
                                      float(dims[2]), float(level_id), _ptr(out_boxes), _ptr(out_scores), _ptr(out_levels),
                                      _stream()), "sis3d_proposal_decode")


def proposal_decode2(a1, d1, p1, in1, lid1, a2, d2, p2, in2, lid2, dims, out_boxes, out_scores, out_levels):
    """both pyramid levels in one launch (sis3d_proposal_decode2): level 1 -> rows [0, n1), level 2 -> rows [n1, n1 + n2)"""
    n1, n2 = int(in1.numel()), int(in2.numel())
    check(lib().sis3d_proposal_decode2(_ptr(a1), _ptr(d1), _ptr(p1), _ptr(in1), n1, float(lid1), _ptr(a2), _ptr(d2), _ptr(p2), _ptr(in2),
                                       n2, float(lid2), float(dims[0]), float(dims[1]), float(dims[2]), _ptr(out_boxes), _ptr(out_scores),
                                       _ptr(out_levels), _stream()), "sis3d_proposal_decode2")


RECORD_WIDTH = 16


def pack_records(d, dims, origin=None, want_block=True, mail=None):
    """detect() output dict -> (records (K,16), block (1+16K) | None): sis3d_pack_records (include/sis3d.h).
    mail (an ops.Mailbox; needs want_block and K <= 256): the launch is also the LAST node of a mailbox pipeline
    (sis3d_pack_records_post): the block goes to the slot's destination row and the slot is consumed -> `d["_mail_posted"]` = True"""
    rois = d["rois"].contiguous()
    K = rois.shape[0]
    rec = torch.empty(K, RECORD_WIDTH, device=rois.device)
    blk = torch.empty(1 + K * RECORD_WIDTH, device=rois.device) if want_block else None
    has = "cls_pred" in d
    NC = d["cls_prob"].shape[1] if has else 0
    if mail is not None and want_block and K <= 256:
        check(lib().sis3d_pack_records_post(_ptr(rois), _ptr(d["scores"].contiguous()), _ptr(d["levels"].contiguous()),
                                            _ptr(d["cls_pred"].contiguous()) if has else None, _ptr(d["cls_prob"].contiguous()) if has else None,
                                            _ptr(d["bbox_pred"].contiguous()) if has else None, _ptr(d["num"]), _ptr(origin), K, NC,
                                            float(dims[0]), float(dims[1]), float(dims[2]), _ptr(rec), _ptr(blk), _ptr(mail.state),
                                            _ptr(mail._progress_t), _stream()), "sis3d_pack_records_post")
        d["_mail_posted"] = True
        return rec, blk
    check(lib().sis3d_pack_records(_ptr(rois), _ptr(d["scores"].contiguous()), _ptr(d["levels"].contiguous()),
                                   _ptr(d["cls_pred"].contiguous()) if has else None, _ptr(d["cls_prob"].contiguous()) if has else None,
                                   _ptr(d["bbox_pred"].contiguous()) if has else None, _ptr(d["num"]), _ptr(origin), K, NC,
                                   float(dims[0]), float(dims[1]), float(dims[2]), _ptr(rec), _ptr(blk), _stream()),
          "sis3d_pack_records")
    return rec, blk


def topk_desc(scores, k):
    """stable descending top-k of a 1-D score vector -> (scores_sorted (k,), order (k,) int64); k <= 1024"""
    scores = _dev(scores, "scores").contiguous()
    n = scores.numel()
    k = min(int(k), n)
    out_s = torch.empty(k, device=scores.device)
    out_i = torch.empty(k, dtype=torch.int64, device=scores.device)
    check(lib().sis3d_topk_desc(_ptr(scores), n, k, _ptr(out_s), _ptr(out_i), _stream()), "sis3d_topk_desc")
    return out_s, out_i


def softmax2(score):
    """F.softmax over dim 1 of a contiguous (1,2,...) tensor (network.py:546)."""
    score = _dev(score, "score")
    if not score.is_contiguous() or score.shape[0] != 1 or score.shape[1] != 2:
        raise _lib.Sis3dError("softmax2 expects a contiguous (1,2,...) tensor")
    prob = torch.empty_like(score)
    check(lib().sis3d_softmax2(_ptr(score), _ptr(prob), score.numel() // 2, _stream()), "sis3d_softmax2")
    return prob


# ----------------------------------------------------------------------- conv --
def new_act(C, dims, device):
    """channels-last activation: logical (1,C,X,Y,Z), memory (X,Y,Z,C)."""
    return torch.empty(tuple(dims) + (C,), device=device).permute(3, 0, 1, 2).unsqueeze(0)


def is_cl(t):
    """True if logical (1,C,X,Y,Z) tensor t is dense channels-last in memory."""
    if t.dim() != 5 or t.shape[0] != 1:
        return False
    _, C, X, Y, Z = t.shape
    return tuple(t.stride()[1:]) == (1, Y * Z * C, Z * C, C)


def to_cl(t):
    """any (1,C,X,Y,Z) float tensor -> channels-last memory (HIP transpose kernel when planar)."""
    if isinstance(t, ProjectedVolume):
        return t
    t = _dev(t, "activation")
    if is_cl(t):
        return t
    _, C, X, Y, Z = t.shape
    if t.is_contiguous():
        out = new_act(C, (X, Y, Z), t.device)
        check(lib().sis3d_planar_to_cl(_ptr(t), C, X * Y * Z, _ptr(out), _stream()), "sis3d_planar_to_cl")
        return out
    return to_cl(t.contiguous())


def to_planar(t):
    t = _dev(t, "activation")
    if t.is_contiguous():
        return t
    if is_cl(t):
        _, C, X, Y, Z = t.shape
        out = torch.empty(t.shape, device=t.device)
        check(lib().sis3d_cl_to_planar(_ptr(t), C, X * Y * Z, _ptr(out), _stream()), "sis3d_cl_to_planar")
        return out
    return t.contiguous()


import os as _os

K3_LEGACY = bool(_os.environ.get("SIS3D_K3_LEGACY"))       # A/B switch: every k3 conv through conv3d.hip's 32x32 tiles
K3_BRICK = int(_os.environ.get("SIS3D_K3_BRICK", "-1"))    # tuning hook: force a brick of sis3d_conv3d_k3t16
MLP_LEGACY = bool(_os.environ.get("SIS3D_MLP_LEGACY"))     # A/B switch: classifier through mlp.hip (32x32 tiles)
PW_LEGACY = bool(_os.environ.get("SIS3D_PW_LEGACY"))       # A/B switch: 1x1x1 convs through conv3d.hip only


def conv3d_pw16(x, pc, residual=None, relu=True, out=None, out_coff=0, stage=None, want_main=True):
    """1x1x1 conv (+ bias, + residual, ReLU) and optionally the next 1x1x1 conv on its result, chained through registers
    (sis3d_conv3d_pw16).  -> (main | None, stage_out | None); raises Sis3dUnsupported for shapes without an instantiation."""
    if pc.packed_pw16 is None or (stage is not None and stage["pc"].packed_pw16 is None):
        raise Sis3dUnsupported("no pw16 pack for this layer")
    if pc.cout % 16 or (stage is not None and stage["pc"].cout % 16):
        # packs with zero rows appended (pad_cout16: the RPN heads, the mask head's last layer) belong to the callers that own a padded output
        raise Sis3dUnsupported("pw16 pack padded to whole cout tiles")
    _, cin_t, X, Y, Z = x.shape
    od = (X, Y, Z)
    if cin_t != pc.cin:
        raise _lib.Sis3dError("conv3d_pw16: activation has %d channels, packed weight expects %d" % (cin_t, pc.cin))
    if want_main:
        if out is None:
            out, out_coff = new_act(pc.cout, od, x.device), 0
        elif not is_cl(out) or tuple(out.shape[2:]) != od or out_coff + pc.cout > out.shape[1]:
            raise _lib.Sis3dError("conv3d_pw16: bad `out`")
    else:
        out = None
    if residual is not None and (not is_cl(residual) or tuple(residual.shape[2:]) != od or residual.shape[1] != pc.cout):
        raise _lib.Sis3dError("conv3d_pw16: residual shape mismatch")
    flags = (EPI_RELU if relu else 0) | (EPI_RESIDUAL if residual is not None else 0)
    so, spc = None, None
    if stage is not None:
        spc = stage["pc"]
        if spc.k != 1 or spc.cin != pc.cout:
            raise _lib.Sis3dError("conv3d_pw16: stage expects %d input channels, k=1" % pc.cout)
        so = new_act(spc.cout, od, x.device)
    rc = lib().sis3d_conv3d_pw16(_ptr(x), X * Y * Z, pc.cin, cin_t, _ptr(pc.packed_pw16), _ptr(pc.bias), pc.cout, flags, _ptr(residual),
                                 residual.shape[1] if residual is not None else 0, _ptr(out), out.shape[1] if out is not None else 0,
                                 int(out_coff), _ptr(spc.packed_pw16) if spc else None, _ptr(spc.bias) if spc else None,
                                 spc.cout if spc else 0, (EPI_RELU if stage.get("relu", True) else 0) if spc else 0, _ptr(so),
                                 spc.cout if spc else 0, _stream())
    if rc == -4:
        raise Sis3dUnsupported("no pw16 instantiation for %d -> %d -> %s" % (pc.cin, pc.cout, spc.cout if spc else None))
    check(rc, "sis3d_conv3d_pw16")
    return out, so


def conv3d_k3t16(xs, pcs, relu=True, outs=None, out_coff=0, brick=None):
    """Conv3d(k3, p1) + bias (+ ReLU) of 1..4 same-shape problems in one launch of the balanced kernel
    (sis3d_conv3d_k3t16).  xs: channels-last activations; pcs: PackedConv with .packed_t16.  -> list of outputs."""
    n = len(xs)
    x0, p0 = xs[0], pcs[0]
    for x, pc in zip(xs, pcs):
        if not is_cl(x) or x.shape != x0.shape or (pc.cin, pc.cout, pc.k) != (p0.cin, p0.cout, 3) or pc.packed_t16 is None \
                or (pc.bias is None) != (p0.bias is None):
            raise _lib.Sis3dError("conv3d_k3t16: problems must share shape and geometry (k3, cin % 32 == 0)")
    _, cin_t, X, Y, Z = x0.shape
    if cin_t != p0.cin:
        raise _lib.Sis3dError("conv3d_k3t16: activation has %d channels, packed weight expects %d" % (cin_t, p0.cin))
    if WINOGRAD and brick is None and all(getattr(pc, "_w", None) is not None for pc in pcs) and \
            lib().sis3d_conv3d_k3wino_prefer(X, Y, Z, p0.cin, p0.cout, n, int(regime()[0])):
        try:
            return conv3d_k3wino(xs, pcs, relu=relu, outs=outs, out_coff=out_coff)
        except Sis3dUnsupported:
            pass
    if outs is None:
        outs, out_coff = [new_act(p0.cout, (X, Y, Z), x0.device) for _ in range(n)], 0
    for o in outs:
        if not is_cl(o) or tuple(o.shape[2:]) != (X, Y, Z) or out_coff + p0.cout > o.shape[1]:
            raise _lib.Sis3dError("conv3d_k3t16: bad `out`")
    arr = ctypes.c_void_p * n
    ins = arr(*[x.data_ptr() for x in xs])
    wps = arr(*[pc.packed_t16.data_ptr() for pc in pcs])
    bs = arr(*[pc.bias.data_ptr() for pc in pcs]) if p0.bias is not None else None
    os_ = arr(*[o.data_ptr() for o in outs])
    if brick is None:
        brick = K3_BRICK
        if brick < 0 and regime()[1] > 0:              # the thread's regime caps the brick: ask for the choice under that cap
            brick = lib().sis3d_conv3d_k3t16_brick(X, Y, Z, p0.cin, p0.cout, n, regime()[1])
    rc = lib().sis3d_conv3d_k3t16(n, ins, X, Y, Z, p0.cin, cin_t, wps, bs, p0.cout, EPI_RELU if relu else 0, os_, outs[0].shape[1],
                                  int(out_coff), int(brick), _stream())
    if rc == -4:
        raise Sis3dUnsupported("conv3d_k3t16: unsupported shape")
    check(rc, "sis3d_conv3d_k3t16")
    return outs


# ---- Winograd F(2x2x2, 3x3x3), exact fp32 (csrc/conv3d_wino.hip): the default route of the k3 convs that conv3d_k3t16 serves
WINOGRAD = bool(int(_os.environ.get("SIS3D_WINOGRAD", "1") or 0))


# bench.py's accounting hook: while a dict is installed, every launch of the Winograd kernel adds the ALGORITHMIC (direct-convolution)
# FLOP count of the problems it serves, so that the bench can price a stage on the FLOPs the matrix pipe really executes
# (algorithmic - saved, saved = wino_algorithmic * (1 - 64/216)) whatever the dispatch rule currently sends to that kernel
_TALLY = _threading.local()         # per thread, like the dispatch regime: a thread counts its own launches


def flop_tally(on):
    """start (True -> returns the fresh dict) / stop (False -> returns the filled dict) the Winograd FLOP accounting of the
    calling thread"""
    old = getattr(_TALLY, "value", None)
    _TALLY.value = {"wino_algorithmic_flops": 0.0, "wino_launches": 0} if on else None
    return _TALLY.value if on else old


def _tally_wino(flops):
    t = getattr(_TALLY, "value", None)
    if t is not None:
        t["wino_algorithmic_flops"] += float(flops)
        t["wino_launches"] += 1


def set_winograd(on):
    """Algorithm switch for the k3 / pad-1 convs that run on conv3d_k3t16: True (default) = layers the Winograd kernel is expected
    to win on (sis3d_conv3d_k3wino_prefer: the two rpn_net_level* convs, 58 % of the network's FLOPs) take Winograd
    F(2x2x2, 3x3x3) in fp32 (3.375x fewer multiplications; only binary32 adds and fp32 MFMAs, same error class as the direct
    kernel), False = every layer on the direct implicit-GEMM kernel.  Read at launch / capture time."""
    global WINOGRAD
    WINOGRAD = bool(on)


def packed_wino(pc):
    """transformed-weight pack U = G g G^T of a k3 PackedConv (sis3d_conv_k3wino_pack_weight), built on first use"""
    if getattr(pc, "_packed_wino", None) is None:
        if getattr(pc, "_w", None) is None:
            raise Sis3dUnsupported("winograd conv: no fp32 weight kept for this layer")
        if torch.cuda.is_current_stream_capturing():
            # the pack is a kernel + an allocation: inside a capture it would be replayed with the graph and live in the graph's pool
            raise _lib.Sis3dError("winograd conv: first use of this layer inside a graph capture; run one eager pass (warm-up) first")
        n = lib().sis3d_conv_k3wino_packed_floats(pc.cout, pc.cin)
        if n == 0:
            raise Sis3dUnsupported("winograd conv: cin % 8 != 0")
        pc._packed_wino = torch.empty(n, device=pc._w.device)
        check(lib().sis3d_conv_k3wino_pack_weight(_ptr(pc._w), pc.cout, pc.cin, _ptr(pc._packed_wino), _stream()), "sis3d_conv_k3wino_pack_weight")
    return pc._packed_wino


def conv3d_k3wino(xs, pcs, relu=True, outs=None, out_coff=0):
    """Conv3d(k3, p1) + bias (+ ReLU) of 1..4 same-shape problems in one launch of the Winograd kernel (sis3d_conv3d_k3wino)."""
    n = len(xs)
    x0, p0 = xs[0], pcs[0]
    for x, pc in zip(xs, pcs):
        if not is_cl(x) or x.shape != x0.shape or (pc.cin, pc.cout, pc.k) != (p0.cin, p0.cout, 3) or (pc.bias is None) != (p0.bias is None):
            raise _lib.Sis3dError("conv3d_k3wino: problems must share shape and geometry (k3)")
    _, cin_t, X, Y, Z = x0.shape
    if cin_t != p0.cin:
        raise _lib.Sis3dError("conv3d_k3wino: activation has %d channels, packed weight expects %d" % (cin_t, p0.cin))
    wps_t = [packed_wino(pc) for pc in pcs]
    if outs is None:
        outs, out_coff = [new_act(p0.cout, (X, Y, Z), x0.device) for _ in range(n)], 0
    for o in outs:
        if not is_cl(o) or tuple(o.shape[2:]) != (X, Y, Z) or out_coff + p0.cout > o.shape[1]:
            raise _lib.Sis3dError("conv3d_k3wino: bad `out`")
    arr = ctypes.c_void_p * n
    ins = arr(*[x.data_ptr() for x in xs])
    wps = arr(*[w.data_ptr() for w in wps_t])
    bs = arr(*[pc.bias.data_ptr() for pc in pcs]) if p0.bias is not None else None
    os_ = arr(*[o.data_ptr() for o in outs])
    flags = (EPI_RELU if relu else 0) | (DISPATCH_SHARED_CHIP if regime()[0] else 0)
    pig, rc = getattr(_REGIME, "piggy", None), -4
    if pig is not None and not pig[2][0] and 2.0 * n * X * Y * Z * p0.cin * p0.cout * 27 >= PIGGY_MIN_FLOPS:
        rc = lib().sis3d_conv3d_k3wino_piggyback(n, ins, X, Y, Z, p0.cin, cin_t, wps, bs, p0.cout, flags, os_, outs[0].shape[1],
                                                 int(out_coff), _ptr(pig[0].state), _ptr(pig[1]), pig[1].numel(), _stream())
        pig[2][0] = rc == 0
    if rc == -4:                                     # no piggyback asked for, or this launch has no room for it
        rc = lib().sis3d_conv3d_k3wino(n, ins, X, Y, Z, p0.cin, cin_t, wps, bs, p0.cout, flags, os_, outs[0].shape[1],
                                       int(out_coff), _stream())
    if rc == -4:
        raise Sis3dUnsupported("conv3d_k3wino: unsupported shape")
    check(rc, "sis3d_conv3d_k3wino")
    _tally_wino(2.0 * n * X * Y * Z * p0.cin * p0.cout * 27)
    return outs


BNECK_WINO = _os.environ.get("SIS3D_BNECK_WINO", "1") != "0"     # A/B switch: planes-32 Bottleneck bodies on the Winograd kernel (default on)
BNECK_SPLIT = bool(_os.environ.get("SIS3D_BNECK_SPLIT"))  # A/B switch: Bottleneck body as k3t16 + pointwise launches


def bottleneck_wino(y1, pc2, pc3, residual, out=None, out_coff=0, stage=None):
    """Bottleneck body in one launch of the Winograd kernel (sis3d_bottleneck_wino): conv2 by F(2x2x2, 3x3x3) in exact fp32, then on
    the output tile relu(conv3(.) + b3 + x) -> `out` and optionally the next block's conv1.  Same arguments and results as
    bottleneck16 (which routes here where sis3d_bottleneck_wino_prefer says so); planes = 32 only -> Sis3dUnsupported otherwise."""
    spc = stage["pc"] if stage is not None else None
    if getattr(pc2, "_w", None) is None or pc3.packed_pw16 is None or (spc is not None and spc.packed_pw16 is None):
        raise Sis3dUnsupported("no Winograd / pw16 pack for this Bottleneck")
    if not is_cl(y1) or not is_cl(residual):
        raise _lib.Sis3dError("bottleneck_wino expects channels-last activations")
    _, pl, X, Y, Z = y1.shape
    od = (X, Y, Z)
    if (pc2.cin, pc2.cout, pc2.k, pc3.cin, pc3.k) != (pl, pl, 3, pl, 1) or tuple(residual.shape[2:]) != od or residual.shape[1] != pc3.cout:
        raise _lib.Sis3dError("bottleneck_wino: layer shapes do not form a Bottleneck")
    if spc is not None and (spc.k != 1 or spc.cin != pc3.cout or not stage.get("relu", True)):
        raise Sis3dUnsupported("bottleneck_wino: stage must be a k=1 conv + ReLU on the block output")
    wp = packed_wino(pc2)
    if out is None:
        out, out_coff = new_act(pc3.cout, od, y1.device), 0
    elif not is_cl(out) or tuple(out.shape[2:]) != od or out_coff + pc3.cout > out.shape[1]:
        raise _lib.Sis3dError("bottleneck_wino: bad `out`")
    so = new_act(spc.cout, od, y1.device) if spc is not None else None
    rc = lib().sis3d_bottleneck_wino(_ptr(y1), X, Y, Z, pl, _ptr(wp), _ptr(pc2.bias), _ptr(pc3.packed_pw16), _ptr(pc3.bias), pc3.cout,
                                     _ptr(residual), residual.shape[1], _ptr(out), out.shape[1], int(out_coff),
                                     _ptr(spc.packed_pw16) if spc else None, _ptr(spc.bias) if spc else None, spc.cout if spc else 0,
                                     _ptr(so), _stream())
    if rc == -4:
        raise Sis3dUnsupported("no Winograd Bottleneck instantiation for planes %d, %d channels, stage %s" % (pl, pc3.cout, spc.cout if spc else 0))
    check(rc, "sis3d_bottleneck_wino")
    _tally_wino(2.0 * X * Y * Z * pl * pl * 27)
    return out, so


def bottleneck16(y1, pc2, pc3, residual, out=None, out_coff=0, stage=None, brick=-1):
    """Bottleneck body in one launch (sis3d_bottleneck16): relu(conv3(relu(conv2(y1))) + x) -> `out`, and optionally the next
    block's conv1 on it.  y1: channels-last conv1 output; pc2: k3 PackedConv (.packed_t16); pc3 / stage['pc']: k1 PackedConv
    (.packed_pw16); residual: the block input.  -> (out, stage_out | None); raises Sis3dUnsupported."""
    spc = stage["pc"] if stage is not None else None
    if BNECK_SPLIT or pc2.packed_t16 is None or pc3.packed_pw16 is None or (spc is not None and spc.packed_pw16 is None):
        raise Sis3dUnsupported("no fused Bottleneck pack")
    if not is_cl(y1) or not is_cl(residual):
        raise _lib.Sis3dError("bottleneck16 expects channels-last activations")
    _, pl, X, Y, Z = y1.shape
    od = (X, Y, Z)
    if (pc2.cin, pc2.cout, pc2.k, pc3.cin, pc3.k) != (pl, pl, 3, pl, 1) or tuple(residual.shape[2:]) != od or residual.shape[1] != pc3.cout:
        raise _lib.Sis3dError("bottleneck16: layer shapes do not form a Bottleneck")
    if spc is not None and (spc.k != 1 or spc.cin != pc3.cout or not stage.get("relu", True)):
        raise Sis3dUnsupported("bottleneck16: stage must be a k=1 conv + ReLU on the block output")
    if WINOGRAD and BNECK_WINO and brick < 0 and lib().sis3d_bottleneck_wino_prefer(X, Y, Z, pl, pc3.cout, spc.cout if spc else 0, int(regime()[0])):
        try:
            # r4: conv2 by Winograd F(2x2x2, 3x3x3) with the 1x1x1 tail in its epilogue: the planes = 32 blocks of the 48 x 24 x 48 maps
            return bottleneck_wino(y1, pc2, pc3, residual, out=out, out_coff=out_coff, stage=stage)
        except Sis3dUnsupported:
            pass
    elif WINOGRAD and BNECK_WINO and brick < 0 and spc is not None and out is None and spc.packed_pw16 is not None and \
            lib().sis3d_bottleneck_wino_prefer(X, Y, Z, pl, pc3.cout, 0, int(regime()[0])):
        # shared chip, Bottleneck(128, 32) of the 24 x 12 x 24 maps: the body on the Winograd kernel (27 work items) and the NEXT block's
        # conv1 as its own pointwise launch -- the kernel has no instantiation that holds both 128-channel weight sets of the tail
        try:
            main, _ = bottleneck_wino(y1, pc2, pc3, residual)
            so, _ = conv3d_pw16(main, spc, relu=True)
            return main, so
        except Sis3dUnsupported:
            pass
    if lib().sis3d_bottleneck16_brick(X, Y, Z, pl) < 0 and brick < 0:
        raise Sis3dUnsupported("bottleneck16: the two-launch path serves this grid")
    if out is None:
        out, out_coff = new_act(pc3.cout, od, y1.device), 0
    elif not is_cl(out) or tuple(out.shape[2:]) != od or out_coff + pc3.cout > out.shape[1]:
        raise _lib.Sis3dError("bottleneck16: bad `out`")
    so = new_act(spc.cout, od, y1.device) if spc is not None else None
    rc = lib().sis3d_bottleneck16(_ptr(y1), X, Y, Z, pl, _ptr(pc2.packed_t16), _ptr(pc2.bias), _ptr(pc3.packed_pw16), _ptr(pc3.bias),
                                  pc3.cout, _ptr(residual), residual.shape[1], _ptr(out), out.shape[1], int(out_coff),
                                  _ptr(spc.packed_pw16) if spc else None, _ptr(spc.bias) if spc else None, spc.cout if spc else 0,
                                  _ptr(so), int(brick), _stream())
    if rc == -4:
        raise Sis3dUnsupported("no fused Bottleneck instantiation for planes %d, %d channels" % (pl, pc3.cout))
    check(rc, "sis3d_bottleneck16")
    return out, so


def conv3d_pw_chain(x, pc, residual=None, relu=True, out=None, out_coff=0, stage=None):
    """1x1x1 conv + bias + residual + ReLU written to `out` (channel offset out_coff), optionally followed by ONE fused
    1x1x1 stage on the on-chip tile (the next Bottleneck's conv1): sis3d_conv3d_pw_chain.
    stage: dict(pc=PackedConv(k=1), relu=bool).  -> (main, stage_out | None); raises Sis3dUnsupported."""
    if not is_cl(x) or pc.k != 1 or x.shape[1] != pc.cin:
        raise _lib.Sis3dError("conv3d_pw_chain expects a channels-last activation and a k=1 PackedConv")
    try:
        return conv3d_pw16(x, pc, residual=residual, relu=relu, out=out, out_coff=out_coff, stage=stage)
    except Sis3dUnsupported:
        pass
    _, cin_t, X, Y, Z = x.shape
    od = (X, Y, Z)
    if out is None:
        out, out_coff = new_act(pc.cout, od, x.device), 0
    elif not is_cl(out) or tuple(out.shape[2:]) != od or out_coff + pc.cout > out.shape[1]:
        raise _lib.Sis3dError("conv3d_pw_chain: bad `out`")
    if residual is not None and (not is_cl(residual) or tuple(residual.shape[2:]) != od or residual.shape[1] != pc.cout):
        raise _lib.Sis3dError("conv3d_pw_chain: residual shape mismatch")
    flags = (EPI_RELU if relu else 0) | (EPI_RESIDUAL if residual is not None else 0)
    arr, so = None, None
    if stage is not None:
        spc = stage["pc"]
        if spc.k != 1 or spc.cin != pc.cout:
            raise _lib.Sis3dError("conv3d_pw_chain: stage expects %d input channels, k=1" % pc.cout)
        so = new_act(spc.cout, od, x.device)
        arr = (_lib.PwStage * 1)()
        arr[0].packed_w = spc.packed.data_ptr()
        arr[0].bias = spc.bias.data_ptr() if spc.bias is not None else None
        arr[0].residual = None
        arr[0].out = so.data_ptr()
        arr[0].cin, arr[0].cout = pc.cout, spc.cout
        arr[0].res_stride, arr[0].out_stride = 0, spc.cout
        arr[0].flags = EPI_RELU if stage.get("relu", True) else 0
    rc = lib().sis3d_conv3d_pw_chain(_ptr(x), X, Y, Z, pc.cin, cin_t, _ptr(pc.packed), _ptr(pc.bias), pc.cout, flags, _ptr(residual),
                                     residual.shape[1] if residual is not None else 0, _ptr(out), out.shape[1], int(out_coff),
                                     1 if stage is not None else 0, arr, _stream())
    if rc == -4:
        raise Sis3dUnsupported("no fused pointwise tiling for this shape")
    check(rc, "sis3d_conv3d_pw_chain")
    return out, so


def conv3d_k2s2_pw16(x, pc, relu=True, stage=None, want_main=True):
    """Conv3d(k2, s2) (+bias, ReLU) [+ the following 1x1x1 conv on its result] as one register-chained launch
    (sis3d_conv3d_k2s2_pw16).  -> (main | None, stage_out | None); raises Sis3dUnsupported."""
    if pc.k != 2 or pc.packed_pw16 is None or (stage is not None and stage["pc"].packed_pw16 is None) or not is_cl(x):
        raise Sis3dUnsupported("no pw16 pack for this k2 s2 layer")
    _, cin_t, X, Y, Z = x.shape
    if cin_t != pc.cin:
        raise _lib.Sis3dError("conv3d_k2s2_pw16: activation has %d channels, packed weight expects %d" % (cin_t, pc.cin))
    od = (X // 2, Y // 2, Z // 2)
    main = new_act(pc.cout, od, x.device) if want_main else None
    so, spc = None, None
    if stage is not None:
        spc = stage["pc"]
        if spc.k != 1 or spc.cin != pc.cout:
            raise _lib.Sis3dError("conv3d_k2s2_pw16: stage expects %d input channels, k=1" % pc.cout)
        so = new_act(spc.cout, od, x.device)
    rc = lib().sis3d_conv3d_k2s2_pw16(_ptr(x), X, Y, Z, pc.cin, cin_t, _ptr(pc.packed_pw16), _ptr(pc.bias), pc.cout,
                                      EPI_RELU if relu else 0, _ptr(main), pc.cout, 0, _ptr(spc.packed_pw16) if spc else None,
                                      _ptr(spc.bias) if spc else None, spc.cout if spc else 0,
                                      (EPI_RELU if stage.get("relu", True) else 0) if spc else 0, _ptr(so), spc.cout if spc else 0, _stream())
    if rc == -4:
        raise Sis3dUnsupported("no k2s2 pw16 instantiation for %d -> %d -> %s" % (pc.cin, pc.cout, spc.cout if spc else None))
    check(rc, "sis3d_conv3d_k2s2_pw16")
    return main, so


def pack_stem_planar2(weight):
    """pw16 pack of a Conv3d(2, C, k2, s2) weight viewed as (C, 16) (column = ci*8 + 4 dx + 2 dy + dz).  The caller owns the
    result (HipConv3d keeps it next to the parameter it was made from)."""
    cout = weight.shape[0]
    w2 = _dev(weight.detach(), "weight").reshape(cout, 16).contiguous()
    pk = torch.empty(lib().sis3d_conv_pw16_packed_floats(cout, 16), device=w2.device)
    check(lib().sis3d_conv_pw16_pack_weight(_ptr(w2), cout, 16, _ptr(pk), _stream()), "sis3d_conv_pw16_pack_weight")
    return pk


def stem_planar2(x, packed, cout, relu=True, stage=None):
    """geometry1[0] (Conv3d(2, C, k2, s2, bias=False) + ReLU on the planar grid) chained into the first Bottleneck's conv1
    (sis3d_conv3d_stem_planar2).  packed: pack_stem_planar2(weight).  -> (x0 channels-last, conv1 output | None); raises
    Sis3dUnsupported."""
    x = _dev(x, "scene")
    if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != 2 or x.stride(4) != 1 or PW_LEGACY:
        raise Sis3dUnsupported("stem_planar2 expects the planar (1,2,X,Y,Z) grid")
    _, _, X, Y, Z = x.shape
    pk = packed
    od = (X // 2, Y // 2, Z // 2)
    out = new_act(cout, od, x.device)
    so, spc = None, None
    if stage is not None:
        spc = stage["pc"]
        if spc.packed_pw16 is None or spc.k != 1 or spc.cin != cout:
            raise Sis3dUnsupported("stem_planar2: stage needs a pw16-packed 1x1x1 conv on %d channels" % cout)
        so = new_act(spc.cout, od, x.device)
    st = x.stride()
    rc = lib().sis3d_conv3d_stem_planar2(_ptr(x), st[1], st[2], st[3], X, Y, Z, _ptr(pk), cout, EPI_RELU if relu else 0, _ptr(out), cout,
                                         _ptr(spc.packed_pw16) if spc else None, _ptr(spc.bias) if spc else None, spc.cout if spc else 0,
                                         (EPI_RELU if stage.get("relu", True) else 0) if spc else 0, _ptr(so), spc.cout if spc else 0,
                                         _stream())
    if rc == -4:
        raise Sis3dUnsupported("no planar stem instantiation for 2 -> %d -> %s" % (cout, spc.cout if spc else None))
    check(rc, "sis3d_conv3d_stem_planar2")
    return out, so


def rpn_heads(r1, pc1, A1, r2, pc2, A2):
    """both RPN heads of both pyramid levels in one launch (sis3d_rpn_heads).  r?: rpn_net outputs (channels-last, 256 ch),
    pc?: PackedConv of the stacked (8A, 256) head matrix.  -> ((score, bbox, prob) level 1, (score, bbox, prob) level 2)"""
    if pc1.packed_pw16 is None or pc2.packed_pw16 is None or not is_cl(r1) or not is_cl(r2) or r1.shape != r2.shape:
        raise Sis3dUnsupported("rpn_heads: needs pw16 packs and two channels-last maps of one shape")
    _, C, X, Y, Z = r1.shape
    od = (X, Y, Z)
    outs = []
    for A in (A1, A2):
        score = torch.empty((1, 2) + od + (A,), device=r1.device)
        outs.append((score, torch.empty((1,) + od + (6 * A,), device=r1.device), torch.empty_like(score)))
    (s1, b1, p1), (s2, b2, p2) = outs
    rc = lib().sis3d_rpn_heads(_ptr(r1), _ptr(pc1.packed_pw16), _ptr(pc1.bias), int(A1), _ptr(s1), _ptr(p1), _ptr(b1),
                               _ptr(r2), _ptr(pc2.packed_pw16), _ptr(pc2.bias), int(A2), _ptr(s2), _ptr(p2), _ptr(b2),
                               X * Y * Z, C, C, _stream())
    if rc == -4:
        raise Sis3dUnsupported("rpn_heads: no instantiation for %d / %d anchors on %d channels" % (A1, A2, C))
    check(rc, "sis3d_rpn_heads")
    return outs[0], outs[1]


class PackedConv:
    """Weights of one nn.Conv3d repacked into MFMA fragment order (sis3d_conv_pack_weight)."""

    def __init__(self, weight, bias=None, cin_pad=None, pad_cout16=False):
        """pad_cout16: also build the pw16 pack when cout is not a multiple of 16 (zero rows appended; the RPN heads)"""
        w = _dev(weight.detach(), "weight").contiguous()
        self.cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
        if not (w.shape[2] == w.shape[3] == w.shape[4]):
            raise _lib.Sis3dError("cubic kernels only")
        self.k = k
        self.cin = cin_pad or ((cin + 7) // 8) * 8           # kernel reads cin padded to a multiple of 8
        if self.cin != cin:
            wp = torch.zeros(self.cout, self.cin, k, k, k, device=w.device)
            wp[:, :cin] = w
            w = wp
        n = lib().sis3d_conv_packed_floats(self.cout, self.cin, k)
        self.packed = torch.empty(n, device=w.device)
        check(lib().sis3d_conv_pack_weight(_ptr(w), self.cout, self.cin, k, _ptr(self.packed), _stream()), "sis3d_conv_pack_weight")
        self.bias = _dev(bias.detach(), "bias").contiguous().clone() if bias is not None else None
        self.version = (weight._version, None if bias is None else bias._version, weight.data_ptr())
        # 1x1x1: second pack for the register-chained pointwise kernels (csrc/pointwise.hip): [cout/16][cin/16][64][4]
        self.packed_pw16, self.bias16 = None, None
        if k == 1 and self.cin % 16 == 0 and (self.cout % 16 == 0 or pad_cout16) and not PW_LEGACY:
            npw = lib().sis3d_conv_pw16_packed_floats(self.cout, self.cin)
            self.packed_pw16 = torch.empty(npw, device=w.device)
            check(lib().sis3d_conv_pw16_pack_weight(_ptr(w), self.cout, self.cin, _ptr(self.packed_pw16), _stream()),
                  "sis3d_conv_pw16_pack_weight")
            if self.cout % 16 and self.bias is not None:          # bias rows of the appended zero couts (sis3d_conv3d_pw16 reads whole tiles)
                self.bias16 = torch.zeros((self.cout + 15) // 16 * 16, device=w.device)
                self.bias16[:self.cout] = self.bias
        # k2 s2: the same register-chained kernel with 8 gathered rows (column index tap*Cin + ci)
        if k == 2 and self.cin % 16 == 0 and self.cout % 16 == 0 and not PW_LEGACY:
            w2 = w.permute(0, 2, 3, 4, 1).reshape(self.cout, 8 * self.cin).contiguous()
            npw = lib().sis3d_conv_pw16_packed_floats(self.cout, 8 * self.cin)
            self.packed_pw16 = torch.empty(npw, device=w.device)
            check(lib().sis3d_conv_pw16_pack_weight(_ptr(w2), self.cout, 8 * self.cin, _ptr(self.packed_pw16), _stream()),
                  "sis3d_conv_pw16_pack_weight")
        # second pack for the balanced k3 kernel (csrc/conv3d_t16.hip): [cout/16][cin/32][4][27][64][2]
        self.packed_t16 = None
        self._w = w if k == 3 else None                       # kept for the Winograd pack (packed_wino)
        if k == 3 and self.cin % 32 == 0 and self.cout % 4 == 0 and not K3_LEGACY:
            nt = lib().sis3d_conv_k3t16_packed_floats(self.cout, self.cin)
            self.packed_t16 = torch.empty(nt, device=w.device)
            check(lib().sis3d_conv_k3t16_pack_weight(_ptr(w), self.cout, self.cin, _ptr(self.packed_t16), _stream()),
                  "sis3d_conv_k3t16_pack_weight")


def conv3d(x, pc, stride=1, relu=False, residual=None, sigmoid=False, out=None, out_coff=0, rpn_anchors=0):
    """x: channels-last (1,Cin,X,Y,Z).  -> channels-last (1,Cout,OX,OY,OZ) (or writes `out` at channel
    offset out_coff).  rpn_anchors=A: returns (score (1,2,X,Y,Z,A), bbox (1,X,Y,Z,6A), prob = softmax(score, 1)) contiguous."""
    if not is_cl(x):
        raise _lib.Sis3dError("conv3d expects a channels-last activation (use ops.to_cl)")
    _, cin_t, X, Y, Z = x.shape
    if cin_t != pc.cin:
        raise _lib.Sis3dError("conv3d: activation has %d channels, packed weight expects %d" % (cin_t, pc.cin))
    if pc.k == 2:
        if stride != 2:
            raise _lib.Sis3dError("k=2 convs are stride 2")
        od = (X // 2, Y // 2, Z // 2)
    else:
        if stride != 1:
            raise _lib.Sis3dError("k=1/3 convs are stride 1")
        od = (X, Y, Z)
    if pc.k == 1 and pc.packed_pw16 is not None and not sigmoid and not rpn_anchors:
        try:
            return conv3d_pw16(x, pc, residual=residual, relu=relu, out=out, out_coff=out_coff)[0]
        except Sis3dUnsupported:
            pass
    if pc.k == 3 and pc.packed_t16 is not None and residual is None and not sigmoid and not rpn_anchors:
        try:
            return conv3d_k3t16([x], [pc], relu=relu, outs=None if out is None else [out], out_coff=out_coff)[0]
        except Sis3dUnsupported:
            pass
    flags = (EPI_RELU if relu else 0) | (EPI_RESIDUAL if residual is not None else 0) | (EPI_SIGMOID if sigmoid else 0)
    out2 = None
    o3_ptr = None
    if rpn_anchors:
        A = rpn_anchors
        flags |= EPI_RPN_HEAD
        score = torch.empty((1, 2) + od + (A,), device=x.device)
        bbox = torch.empty((1,) + od + (6 * A,), device=x.device)
        prob = torch.empty_like(score)
        o_ptr, o2_ptr, ostride = _ptr(score), _ptr(bbox), 0
        o3_ptr = _ptr(prob)
        ret = (score, bbox, prob)
    else:
        if out is None:
            out = new_act(pc.cout, od, x.device)
        elif not is_cl(out) or tuple(out.shape[2:]) != od:
            raise _lib.Sis3dError("conv3d: bad `out`")
        o_ptr, o2_ptr, ostride = _ptr(out), None, out.shape[1]
        ret = out
    res_stride = 0
    if residual is not None:
        if not is_cl(residual) or tuple(residual.shape[2:]) != od:
            raise _lib.Sis3dError("conv3d: residual must be channels-last with the output's grid")
        res_stride = residual.shape[1]
    check(lib().sis3d_conv3d(_ptr(x), X, Y, Z, pc.cin, cin_t, _ptr(pc.packed), _ptr(pc.bias), pc.cout, pc.k, stride, flags,
                             _ptr(residual), res_stride, o_ptr, ostride, out_coff, o2_ptr, o3_ptr, rpn_anchors, _stream()), "sis3d_conv3d")
    return ret


# ---- sparse colour stem (csrc/proj_sparse.hip): only the output voxels that see a visible input voxel are computed
SPARSE_PROJECTION = _os.environ.get("SIS3D_SPARSE_PROJECTION", "1") != "0"


def set_sparse_projection(on):
    """True (default): Conv3d(128, 64, k=2, s=2) on a ProjectedVolume runs on the sparse kernels (sis3d_conv3d_k2s2_projected_sparse);
    False: the dense kernel that reads every voxel through the table (sis3d_conv3d_chain_projected).  Same arithmetic, different
    summation order (~1e-6 of the output scale)."""
    global SPARSE_PROJECTION
    SPARSE_PROJECTION = bool(on)


_SPARSE_WS = {}                 # (grid, device, stream handle) -> (weakref to the torch stream object, workspace)
_SPARSE_WS_MAX = 16             # bounded: engines use a handful of (grid, stream) pairs; the oldest entry goes first


def _conv3d_chain_projected_sparse(x, pc, stages, relu, want_main):
    """-> (main, [stage out]) or None when the shape is not the colour stem's (the dense kernel serves it then)"""
    if pc.cin != 128 or pc.cout != 64 or pc.packed_pw16 is None or len(stages) > 1 or any(d % 2 for d in x.dims):
        return None
    spc = None
    if stages:
        st = stages[0]
        spc = st["pc"]
        if (spc.k != 1 or spc.cin != 64 or spc.cout != 32 or spc.packed_pw16 is None or not st.get("relu", True) or st.get("residual") is not None
                or st.get("out") is not None):
            return None
    X, Y, Z = x.dims
    od = (X // 2, Y // 2, Z // 2)
    # the C entry's own limits (csrc/proj_sparse.hip: MAXSL view slots, row index in 24 bits), checked BEFORE anything is allocated
    if x.nslots > 6 or od[0] * od[1] * od[2] >= (1 << 24):
        return None
    main = new_act(pc.cout, od, x.device)
    so = new_act(spc.cout, od, x.device) if spc is not None else None
    # eager launches share one workspace per (grid, device, stream) -- launches on a stream are ordered; a capture gets its own from
    # the graph's pool, because graphs captured on one stream may replay concurrently on several
    capturing = torch.cuda.is_current_stream_capturing()
    key = (X, Y, Z, x.device, int(torch.cuda.current_stream().cuda_stream))
    ws = None if capturing else _SPARSE_WS.pop(key, None)
    if ws is None:
        ws = torch.empty(lib().sis3d_conv3d_k2s2_projected_sparse_workspace_bytes(X, Y, Z), dtype=torch.uint8, device=x.device)
    if not capturing:
        # most recently used last; bounded.  A recycled stream handle inheriting an old workspace is harmless: launches on one
        # stream are ordered, and the buffer belongs to the caching allocator's pool of the stream it was allocated on (the
        # pipelines' streams of sis3d.engine live as long as the process: engine.pooled_stream)
        _SPARSE_WS[key] = ws
        while len(_SPARSE_WS) > _SPARSE_WS_MAX:
            _SPARSE_WS.pop(next(iter(_SPARSE_WS)))
    wsb = ws.numel()
    rc = lib().sis3d_conv3d_k2s2_projected_sparse(_ptr(x.table), _ptr(x.rows), x.nslots, x.npix, X, Y, Z, pc.cin, _ptr(pc.packed_pw16),
                                                  _ptr(pc.bias), pc.cout, 1 if relu else 0, _ptr(main), _ptr(spc.packed_pw16) if spc else None,
                                                  _ptr(spc.bias) if spc else None, spc.cout if spc else 0, _ptr(so), _ptr(ws), wsb, _stream())
    if rc == -4:
        return None
    check(rc, "sis3d_conv3d_k2s2_projected_sparse")
    return (main if want_main else None), ([so] if spc is not None else [])


def conv3d_chain(x, pc, stride, stages, relu=True, want_main=False):
    """Main conv (k3 / k2s2, + bias, ReLU) followed by fused 1x1x1 stages on the on-chip tile (sis3d_conv3d_chain).
    stages: list of dicts(pc=PackedConv(k=1), relu=bool, residual=tensor|None, keep=bool).  Returns
    (main_out | None, [stage outputs | None]).  Raises Sis3dUnsupported if no tiling can fuse this shape."""
    proj = isinstance(x, ProjectedVolume)
    if proj and (pc.k != 2 or stride != 2 or pc.cin != x.C):
        raise Sis3dUnsupported("a projected volume feeds Conv3d(C, *, k=2, s=2) only")
    if proj and SPARSE_PROJECTION:
        r = _conv3d_chain_projected_sparse(x, pc, stages, relu, want_main)
        if r is not None:
            return r
    if not proj and not is_cl(x):
        raise _lib.Sis3dError("conv3d_chain expects a channels-last activation")
    _, cin_t, X, Y, Z = x.shape
    od = (X // 2, Y // 2, Z // 2) if pc.k == 2 else (X, Y, Z)
    main = new_act(pc.cout, od, x.device) if want_main else None
    arr = (_lib.PwStage * len(stages))()
    outs = []
    cprev = pc.cout
    for i, st in enumerate(stages):
        spc = st["pc"]
        last = i + 1 == len(stages)
        ostride, optr = spc.cout, None
        if st.get("out") is not None:                     # write into a channel range of a wider tensor (torch.cat fusion)
            o = st["out"]
            if not is_cl(o) or tuple(o.shape[2:]) != od:
                raise _lib.Sis3dError("conv3d_chain: bad stage `out`")
            ostride, optr = o.shape[1], o.data_ptr() + 4 * int(st.get("out_coff", 0))
        else:
            o = new_act(spc.cout, od, x.device) if (st.get("keep", True) or last) else None
            optr = o.data_ptr() if o is not None else None
        outs.append(o)
        res = st.get("residual")
        if res is not None and (not is_cl(res) or tuple(res.shape[2:]) != od or res.shape[1] != spc.cout):
            raise _lib.Sis3dError("conv3d_chain: residual shape mismatch")
        arr[i].packed_w = spc.packed.data_ptr()
        arr[i].bias = spc.bias.data_ptr() if spc.bias is not None else None
        arr[i].residual = res.data_ptr() if res is not None else None
        arr[i].out = optr
        arr[i].cin, arr[i].cout = cprev, spc.cout
        arr[i].res_stride = res.shape[1] if res is not None else 0
        arr[i].out_stride = ostride
        arr[i].flags = (EPI_RELU if st.get("relu", True) else 0) | (EPI_RESIDUAL if res is not None else 0)
        if spc.cin != cprev or spc.k != 1:
            raise _lib.Sis3dError("conv3d_chain: stage %d expects %d input channels, k=1" % (i, cprev))
        cprev = spc.cout
    if proj:
        rc = lib().sis3d_conv3d_chain_projected(_ptr(x.table), _ptr(x.rows), x.nslots, x.npix, X, Y, Z, pc.cin, _ptr(pc.packed),
                                                _ptr(pc.bias), pc.cout, EPI_RELU if relu else 0, _ptr(main), pc.cout, len(stages),
                                                arr, _stream())
    else:
        rc = lib().sis3d_conv3d_chain(_ptr(x), X, Y, Z, pc.cin, cin_t, _ptr(pc.packed), _ptr(pc.bias), pc.cout, pc.k, stride,
                                      EPI_RELU if relu else 0, _ptr(main), pc.cout, len(stages), arr, _stream())
    if rc == -4:
        raise Sis3dUnsupported("no fused tiling for this shape")
    check(rc, "sis3d_conv3d_chain")
    return main, outs


class Sis3dUnsupported(_lib.Sis3dError):
    pass


def conv3d_batched(xs, pcs, stride=1, relu=False, residuals=None):
    """Same-shape independent convolutions in one launch (sis3d_conv3d_batched).  xs: channels-last activations
    of identical shape; pcs: PackedConv objects of identical geometry.  -> list of outputs."""
    n = len(xs)
    x0, p0 = xs[0], pcs[0]
    if p0.k == 3 and stride == 1 and residuals is None and all(pc.packed_t16 is not None for pc in pcs):
        try:
            return conv3d_k3t16(xs, pcs, relu=relu)
        except Sis3dUnsupported:
            pass
    for x, pc in zip(xs, pcs):
        if not is_cl(x) or x.shape != x0.shape or (pc.cin, pc.cout, pc.k) != (p0.cin, p0.cout, p0.k) or (pc.bias is None) != (p0.bias is None):
            raise _lib.Sis3dError("conv3d_batched: problems must share shape and geometry")
    _, cin_t, X, Y, Z = x0.shape
    od = (X // 2, Y // 2, Z // 2) if p0.k == 2 else (X, Y, Z)
    outs = [new_act(p0.cout, od, x0.device) for _ in range(n)]
    flags = (EPI_RELU if relu else 0) | (EPI_RESIDUAL if residuals is not None else 0)
    arr = ctypes.c_void_p * n
    ins = arr(*[x.data_ptr() for x in xs])
    wps = arr(*[pc.packed.data_ptr() for pc in pcs])
    bs = arr(*[pc.bias.data_ptr() for pc in pcs]) if p0.bias is not None else None
    rs = arr(*[r.data_ptr() for r in residuals]) if residuals is not None else None
    os_ = arr(*[o.data_ptr() for o in outs])
    check(lib().sis3d_conv3d_batched(n, ins, X, Y, Z, p0.cin, cin_t, wps, bs, p0.cout, p0.k, stride, flags, rs,
                                     residuals[0].shape[1] if residuals is not None else 0, os_, p0.cout, 0, _stream()),
          "sis3d_conv3d_batched")
    return outs


def conv3d_planar2(x, weight, ksize, relu=True, window=None, cout_stride=None):
    """First layers on the planar 2-channel grid (geometry1.0 k2s2; mask conv0 k3p1 on a crop window).
    x: (1,2,X,Y,Z) with a contiguous z axis; weight: checkpoint layout (Cout,2,k,k,k)."""
    x = _dev(x, "scene")
    w = _dev(weight.detach(), "weight").contiguous()
    if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != 2 or x.stride(4) != 1:
        raise _lib.Sis3dError("planar2 conv expects (1,2,X,Y,Z) with contiguous z")
    _, _, X, Y, Z = x.shape
    S = 2 if ksize == 2 else 1
    if window is None:
        window = (0, 0, 0, X // S * S, Y // S * S, Z // S * S)
    x0, y0, z0, x1, y1, z1 = window
    od = ((x1 - x0) // S, (y1 - y0) // S, (z1 - z0) // S)
    cout = w.shape[0]
    cs = cout_stride or cout
    out = new_act(cs, od, x.device)
    if cs != cout:
        out.zero_()
    st = x.stride()
    check(lib().sis3d_conv3d_planar2(_ptr(x), st[1], st[2], st[3], X, Y, Z, x0, y0, z0, od[0], od[1], od[2], _ptr(w), cout, ksize,
                                     EPI_RELU if relu else 0, _ptr(out), cs, _stream()), "sis3d_conv3d_planar2")
    return out


class PackedClassifier:
    """classifier MLP + heads packed for sis3d_classifier_forward.  fc0's columns are permuted from the reference's
    (C, bins) flatten order to the (bins, C) memory order of the channels-last pooled features."""

    def __init__(self, fcs, cls_head, box_head, pool_c, pool_bins):
        (w1, b1), (w2, b2), (w3, b3) = [(m.weight.detach(), m.bias.detach()) for m in fcs]
        w1 = w1.view(w1.shape[0], pool_c, pool_bins).permute(0, 2, 1).reshape(w1.shape[0], -1).contiguous()
        mk = lambda w, b: PackedConv(w.reshape(w.shape[0], w.shape[1], 1, 1, 1), b, pad_cout16=True)
        self.l1, self.l2, self.l3 = mk(w1, b1), mk(w2, b2), mk(w3, b3)
        wh = torch.cat([cls_head.weight.detach(), box_head.weight.detach()], 0)
        bh = torch.cat([cls_head.bias.detach(), box_head.bias.detach()], 0)
        self.head = mk(wh, bh)
        self.nc = cls_head.weight.shape[0]
        self.version = tuple(p._version for m in list(fcs) + [cls_head, box_head] for p in (m.weight, m.bias)) + (fcs[0].weight.data_ptr(),)


def classifier_forward(x, pk, num=None):
    """x (R, K) fp32 rows on the GPU -> (cls_score (R,NC), cls_pred (R,) int64, cls_prob (R,NC), bbox_pred (R,6NC))"""
    x = _dev(x, "pool5")
    if x.dim() != 2 or x.stride(1) != 1:
        raise _lib.Sis3dError("classifier_forward expects (R,K) rows")
    R, K = x.shape
    nc = pk.nc
    dev = x.device
    cls_score = torch.empty(R, nc, device=dev)
    cls_prob = torch.empty(R, nc, device=dev)
    cls_pred = torch.empty(R, dtype=torch.int64, device=dev)
    bbox_pred = torch.empty(R, 6 * nc, device=dev)
    if all(l.packed_pw16 is not None for l in (pk.l1, pk.l2, pk.l3, pk.head)) and not MLP_LEGACY:
        nws = lib().sis3d_classifier16_workspace_floats(R, K, pk.l1.cout)
        ws = torch.empty(max(nws, 1), device=dev)
        rc = lib().sis3d_classifier16_forward(_ptr(x), R, _ptr(num), K, x.stride(0), _ptr(pk.l1.packed_pw16), _ptr(pk.l1.bias), pk.l1.cout,
                                              _ptr(pk.l2.packed_pw16), _ptr(pk.l2.bias), pk.l2.cout, _ptr(pk.l3.packed_pw16),
                                              _ptr(pk.l3.bias), pk.l3.cout, _ptr(pk.head.packed_pw16), _ptr(pk.head.bias), nc,
                                              _ptr(cls_score), _ptr(cls_prob), _ptr(cls_pred), _ptr(bbox_pred), _ptr(ws), nws, _stream())
        if rc != -4:
            check(rc, "sis3d_classifier16_forward")
            return cls_score, cls_pred, cls_prob, bbox_pred
    nws = lib().sis3d_classifier_workspace_floats(R, K, pk.l1.cout)
    ws = torch.empty(max(nws, 1), device=dev)
    check(lib().sis3d_classifier_forward_n(_ptr(x), R, _ptr(num), K, x.stride(0), _ptr(pk.l1.packed), _ptr(pk.l1.bias), pk.l1.cout,
                                         _ptr(pk.l2.packed), _ptr(pk.l2.bias), pk.l2.cout, _ptr(pk.l3.packed), _ptr(pk.l3.bias),
                                         pk.l3.cout, _ptr(pk.head.packed), _ptr(pk.head.bias), nc, _ptr(cls_score), _ptr(cls_prob),
                                           _ptr(cls_pred), _ptr(bbox_pred), _ptr(ws), nws, _stream()), "sis3d_classifier_forward_n")
    return cls_score, cls_pred, cls_prob, bbox_pred


_RAG_DT = None


def _rag_dtypes():
    global _RAG_DT
    if _RAG_DT is None:
        import numpy as np
        _RAG_DT = (np.dtype([("X", "i4"), ("Y", "i4"), ("Z", "i4"), ("nbx", "i4"), ("nby", "i4"), ("nbz", "i4"), ("block0", "i4"),
                             ("pad", "i4"), ("in_off", "i8"), ("out_off", "i8")]),
                   np.dtype([("x0", "i4"), ("y0", "i4"), ("z0", "i4"), ("dx", "i4"), ("dy", "i4"), ("dz", "i4"), ("p0", "i4"),
                             ("p1", "i4"), ("t0", "i8"), ("out_off", "i8")]))
    return _RAG_DT


MASK_MINI = _os.environ.get("SIS3D_MASK_MINI", "1") != "0"     # A/B switch: ragged Winograd launches on MINI geometry (default on)


class MaskPlan:
    """Everything of a ragged mask-head batch that depends only on the crop windows: the three descriptor tables (one
    upload), the packed activation buffers and the per-box output views.  Built on the host once; `mask_head_run` then
    only enqueues kernels, so a plan made BEFORE a graph capture lets the whole mask head be captured (fixed windows)."""

    def __init__(self, windows, C, NC, device):
        import numpy as np
        n = len(windows)
        self.n, self.C, self.NC = n, C, NC
        bx, by, bz, ng = (ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int())
        check(lib().sis3d_ragged_tiling(C, C, 3, ctypes.byref(bx), ctypes.byref(by), ctypes.byref(bz), ctypes.byref(ng)), "sis3d_ragged_tiling")
        bx, by, bz, ng = bx.value, by.value, bz.value, ng.value
        rdt, pdt = _rag_dtypes()
        w = np.asarray(windows, dtype=np.int64).reshape(n, 6)
        ext = w[:, 3:] - w[:, :3]                                     # (n,3) crop sizes
        nbk = -(-ext // np.array([bx, by, bz]))                       # bricks per axis
        nvox = ext.prod(1)
        voffs = np.concatenate([[0], np.cumsum(nvox)])                # voxel offset of every crop in the packed buffers
        blks = np.concatenate([[0], np.cumsum(nbk.prod(1) * ng)])
        d3 = np.zeros(n, dtype=rdt)
        d1 = np.zeros(n, dtype=rdt)
        dp = np.zeros(n, dtype=pdt)
        for d, ostr in ((d3, C), (d1, NC)):
            d["X"], d["Y"], d["Z"] = ext[:, 0], ext[:, 1], ext[:, 2]
            d["nbx"], d["nby"], d["nbz"] = nbk[:, 0], nbk[:, 1], nbk[:, 2]
            d["block0"] = blks[:-1]
            d["in_off"] = voffs[:-1] * C
            d["out_off"] = voffs[:-1] * ostr
        # the k3 layers on the balanced kernel (csrc/conv3d_t16.hip): its own brick / workgroup numbering
        self.t16, self.brick_t16, self.blocks_t16 = False, -1, 0
        d3t = np.zeros(0, dtype=rdt)
        if not K3_LEGACY:
            best = None
            forced = int(_os.environ.get("SIS3D_MASK_BRICK", "-1"))        # tuning hook
            # 3x6x6, 4x4x4, 4x4x8, 3x3x6: least estimated SIMD time wins (6x6x6 = brick 1 exists but measured slower on 9-20 voxel crops:
            # 0.58 vs 0.48 ms for the 16-box batch, measured in r3)
            for brick in ((forced,) if forced >= 0 else (2, 4, 5, 3)):
                tb = [ctypes.c_int() for _ in range(5)]
                if lib().sis3d_ragged_tiling_k3t16(C, C, brick, *[ctypes.byref(v) for v in tb]) != 0:
                    continue
                tbx, tby, tbz, tng, tmt = (v.value for v in tb)
                bdim = np.array([tbx, tby, tbz])
                nbt = -(-ext // bdim)
                # a brick that sticks out of its crop runs only the tile groups it has voxels for (CLIP kernels, conv3d_t16.hip):
                # per workgroup MFMA issue of ceil(ceil(v / 16) / G) * G tiles + a fixed prologue / epilogue share
                grp = 3 if tmt % 3 == 0 else (4 if tmt >= 4 else tmt)
                cost = 0
                for e in ext:
                    per_axis = [np.minimum(bdim[k], e[k] - bdim[k] * np.arange(-(-e[k] // bdim[k]))) for k in range(3)]
                    vox = per_axis[0][:, None, None] * per_axis[1][None, :, None] * per_axis[2][None, None, :]
                    tiles = np.minimum(tmt, -(-(-(-vox // 16)) // grp) * grp)
                    cost += int((tiles * (27 * (C // 16) * 32) + 3000).sum())
                if best is None or cost < best[0]:
                    best = (cost, brick, nbt, tng)
            if best is not None:
                _, brick, nbt, tng = best
                blt = np.concatenate([[0], np.cumsum(nbt.prod(1) * tng)])
                d3t = np.zeros(n, dtype=rdt)
                d3t["X"], d3t["Y"], d3t["Z"] = ext[:, 0], ext[:, 1], ext[:, 2]
                d3t["nbx"], d3t["nby"], d3t["nbz"] = nbt[:, 0], nbt[:, 1], nbt[:, 2]
                d3t["block0"] = blt[:-1]
                d3t["in_off"] = voffs[:-1] * C
                d3t["out_off"] = voffs[:-1] * C
                self.t16, self.brick_t16, self.blocks_t16 = True, brick, int(blt[-1])
        # the same layers on the Winograd kernel (csrc/conv3d_wino.hip): 8x4x8 blocks x groups of two cout tiles; taken when the batch
        # has enough work items to fill the chip (a lone small box stays on the direct kernel)
        self.wino, self.blocks_wino = False, 0
        d3w = np.zeros(0, dtype=rdt)
        tb = [ctypes.c_int() for _ in range(4)]
        if C % 8 == 0 and lib().sis3d_ragged_tiling_k3wino(C, C, *[ctypes.byref(v) for v in tb]) == 0:
            wbx, wby, wbz, wng = (v.value for v in tb)
            nbw = -(-ext // np.array([wbx, wby, wbz]))
            blw = np.concatenate([[0], np.cumsum(nbw.prod(1) * wng)])
            d3w = np.zeros(n, dtype=rdt)
            d3w["X"], d3w["Y"], d3w["Z"] = ext[:, 0], ext[:, 1], ext[:, 2]
            d3w["nbx"], d3w["nby"], d3w["nbz"] = nbw[:, 0], nbw[:, 1], nbw[:, 2]
            d3w["block0"] = blw[:-1]
            d3w["in_off"] = voffs[:-1] * C
            d3w["out_off"] = voffs[:-1] * C
            self.blocks_wino = int(blw[-1])
            self.wino = self.blocks_wino >= int(_os.environ.get("SIS3D_MASK_WINO_MIN", "200"))
        # r4: the Winograd kernel on MINI geometry (work item = four 4 x 4 x 4 minis x two cout tiles): the crops are covered with
        # 4-voxel granularity on every axis instead of 8 x 4 x 8 blocks -- fewer work items for the same boxes (924 minis = 231 quads
        # against 304 blocks on the 16-box bench set: two rounds of the chip per layer instead of three)
        self.wino_mini, self.items_mini = False, 0
        d3m = np.zeros(0, dtype=rdt)
        tm = [ctypes.c_int() for _ in range(3)]
        if C % 8 == 0 and MASK_MINI and lib().sis3d_ragged_tiling_k3wino_mini(C, C, *[ctypes.byref(v) for v in tm]) == 0:
            edge, per_item, mng = (v.value for v in tm)
            nbm = -(-ext // edge)                                         # minis per axis
            quads = -(-nbm.prod(1) // per_item)
            blm = np.concatenate([[0], np.cumsum(quads * mng)])
            d3m = np.zeros(n, dtype=rdt)
            d3m["X"], d3m["Y"], d3m["Z"] = ext[:, 0], ext[:, 1], ext[:, 2]
            d3m["nbx"], d3m["nby"], d3m["nbz"] = nbm[:, 0], nbm[:, 1], nbm[:, 2]
            d3m["block0"] = blm[:-1]
            d3m["in_off"] = voffs[:-1] * C
            d3m["out_off"] = voffs[:-1] * C
            self.items_mini = int(blm[-1])
            self.wino_mini = self.wino and self.items_mini < self.blocks_wino
        dp["x0"], dp["y0"], dp["z0"] = w[:, 0], w[:, 1], w[:, 2]
        dp["dx"], dp["dy"], dp["dz"] = ext[:, 0], ext[:, 1], ext[:, 2]
        dp["t0"] = voffs[:-1] * (C // 4)
        dp["out_off"] = voffs[:-1] * C
        self.voxels, self.blocks, self.items = int(voffs[-1]), int(blks[-1]), int(voffs[-1]) * (C // 4)
        self.dims = [tuple(int(v) for v in e) for e in ext]
        self.windows = [tuple(int(v) for v in r) for r in w]
        # ONE upload for the three descriptor tables (each is a blocking pageable copy)
        parts = [d3.view(np.uint8).reshape(-1), d1.view(np.uint8).reshape(-1), dp.view(np.uint8).reshape(-1),
                 d3t.view(np.uint8).reshape(-1), np.zeros(0, dtype=np.uint8), d3w.view(np.uint8).reshape(-1),
                 d3m.view(np.uint8).reshape(-1)]
        pad = [(-p.size) % 16 for p in parts]
        host = np.concatenate([np.concatenate([p, np.zeros(q, np.uint8)]) for p, q in zip(parts, pad)])
        self.devbuf = torch.from_numpy(host).to(device)
        o1 = parts[0].size + pad[0]
        o2 = o1 + parts[1].size + pad[1]
        o3 = o2 + parts[2].size + pad[2]
        o4 = o3 + parts[3].size + pad[3]
        self.g3, self.g1, self.gp = self.devbuf[:o1], self.devbuf[o1:o2], self.devbuf[o2:o3]
        o5 = o4 + parts[4].size + pad[4]
        o6 = o5 + parts[5].size + pad[5]
        self.g3t, self.g3w, self.g3m = self.devbuf[o3:o4], self.devbuf[o5:o6], self.devbuf[o6:]       # (part 4 was the split-bf16 table, removed in r6)
        self.a = torch.empty(self.voxels, C, device=device)
        self.b = torch.empty(self.voxels, C, device=device)
        # the last layer's rows padded to whole 16-cout tiles (the pointwise kernel stores 16 B per lane); `out` = the (voxels, NC) view
        # (sis3d_conv3d_pw16 is instantiated for 64 / 128 -> 32 couts: 17..32 classes; other heads keep the dense rows of the generic kernel)
        ncp = (NC + 15) // 16 * 16
        self.out_pad = torch.empty(self.voxels, ncp, device=device) if (ncp == 32 and C in (64, 128) and not PW_LEGACY) else None
        self.out = self.out_pad[:, :NC] if self.out_pad is not None else torch.empty(self.voxels, NC, device=device)
        # 2 FLOP per MAC: conv0 (2 -> C, k3), four C -> C k3 convs, the C -> NC k1 head
        self.flops = 2.0 * self.voxels * (54 * C + 4 * 27 * C * C + C * NC)

    def views(self):
        res, voff = [], 0
        for dx, dy, dz in self.dims:
            nv = dx * dy * dz
            res.append(self.out[voff:voff + nv].view(dx, dy, dz, self.NC).permute(3, 0, 1, 2).unsqueeze(0))
            voff += nv
        return res


def mask_head_run(scene, plan, w0, pcs, pc_last, sigmoid=True):
    """enqueue the six ragged launches of a planned mask-head batch; no host work besides the launches"""
    scene = _dev(scene, "scene")
    if scene.dim() != 5 or scene.shape[0] != 1 or scene.shape[1] != 2 or scene.stride(4) != 1:
        raise _lib.Sis3dError("mask_head_batched expects the planar (1,2,X,Y,Z) grid")
    n, C, NC = plan.n, plan.C, plan.NC
    st = scene.stride()
    check(lib().sis3d_conv3d_planar2_ragged(_ptr(scene), st[1], st[2], st[3], _ptr(plan.gp), n, plan.items,
                                            _ptr(_dev(w0.detach(), "w0").contiguous()), C, EPI_RELU, _ptr(plan.a), C, _stream()),
          "sis3d_conv3d_planar2_ragged")
    src, dst = plan.a, plan.b
    for pc in pcs:
        if WINOGRAD and plan.wino_mini and getattr(pc, "_w", None) is not None:
            check(lib().sis3d_conv3d_k3wino_ragged_mini(_ptr(src), C, C, _ptr(packed_wino(pc)), _ptr(pc.bias), C, EPI_RELU, _ptr(dst), C,
                                                        _ptr(plan.g3m), n, plan.items_mini, _stream()), "sis3d_conv3d_k3wino_ragged_mini")
            _tally_wino(2.0 * plan.voxels * C * C * 27)
        elif WINOGRAD and plan.wino and getattr(pc, "_w", None) is not None:
            check(lib().sis3d_conv3d_k3wino_ragged(_ptr(src), C, C, _ptr(packed_wino(pc)), _ptr(pc.bias), C, EPI_RELU, _ptr(dst), C,
                                                   _ptr(plan.g3w), n, plan.blocks_wino, _stream()), "sis3d_conv3d_k3wino_ragged")
            _tally_wino(2.0 * plan.voxels * C * C * 27)
        elif plan.t16 and pc.packed_t16 is not None:
            check(lib().sis3d_conv3d_k3t16_ragged(_ptr(src), C, C, _ptr(pc.packed_t16), _ptr(pc.bias), C, EPI_RELU, _ptr(dst), C,
                                                  _ptr(plan.g3t), n, plan.blocks_t16, plan.brick_t16, _stream()), "sis3d_conv3d_k3t16_ragged")
        else:
            check(lib().sis3d_conv3d_ragged(_ptr(src), C, C, _ptr(pc.packed), _ptr(pc.bias), C, 3, EPI_RELU, _ptr(dst), C, _ptr(plan.g3), n,
                                            plan.blocks, _stream()), "sis3d_conv3d_ragged")
        src, dst = dst, src
    if plan.out_pad is not None and pc_last.packed_pw16 is not None:
        # r6: a 1x1x1 conv does not see the crop structure -- the packed crops are one list of voxel rows for the register-chained
        # pointwise kernel (csrc/pointwise.hip; 11.6 -> 6.2 us for the 16-box batch); its rows are padded to whole 16-cout tiles
        # (MaskPlan.out_pad), plan.out is the (voxels, NC) view of them
        NCP = plan.out_pad.shape[1]
        check(lib().sis3d_conv3d_pw16(_ptr(src), plan.voxels, C, C, _ptr(pc_last.packed_pw16), _ptr(pc_last.bias16 if pc_last.cout % 16 else pc_last.bias),
                                      NCP, EPI_SIGMOID if sigmoid else 0, None, 0, _ptr(plan.out_pad), NCP, 0, None, None, 0, 0, None, 0, _stream()),
              "sis3d_conv3d_pw16")
    elif NC <= 32:
        if plan.out_pad is not None:                               # a plan built for the pointwise kernel, weights without its pack
            raise _lib.Sis3dError("mask head: the last layer has no pw16 pack (SIS3D_PW_LEGACY set after the plan was built?)")
        check(lib().sis3d_conv3d_ragged(_ptr(src), C, C, _ptr(pc_last.packed), _ptr(pc_last.bias), NC, 1, EPI_SIGMOID if sigmoid else 0,
                                        _ptr(plan.out), NC, _ptr(plan.g1), n, plan.blocks, _stream()), "sis3d_conv3d_ragged")
    else:
        # a 1x1x1 conv does not see the crop structure: the packed buffer is one (voxels x 1 x 1) channels-last activation
        # (the 64-feature output of the geometry stack under MASK_USE_IMAGES, lib/nets/backbones.py:246)
        V = plan.voxels
        conv3d(src.view(1, V, 1, 1, C).permute(0, 4, 1, 2, 3), pc_last, sigmoid=sigmoid, out=plan.out.view(1, V, 1, 1, NC).permute(0, 4, 1, 2, 3))
    return plan.views()


class RaggedBatch:
    """A ragged batch of crops (one per detected box) packed back to back, crop after crop, voxels x-major, channels last --
    the packing of MaskPlan -- with descriptor tables built on demand per layer shape for sis3d_conv3d_ragged.  Serves the layers of
    the mask head's colour variants (lib/nets/backbones.py:253-284) that MaskPlan's fixed 64-channel tables do not cover."""

    def __init__(self, windows, device):
        import numpy as np
        self.n = len(windows)
        self.device = device
        w = np.asarray(windows, dtype=np.int64).reshape(self.n, 6)
        self.w = w
        self.ext = w[:, 3:] - w[:, :3]
        self.voffs = np.concatenate([[0], np.cumsum(self.ext.prod(1))])
        self.voxels = int(self.voffs[-1])
        self._tables = {}

    T16_BRICK = 2                                                   # 3x6x6: the brick MaskPlan mostly picks for 9-20 voxel crops

    def table(self, cin, cout, in_stride, out_stride):
        """descriptors of a k3 layer for sis3d_conv3d_k3t16_ragged (block0 counts bricks x cout tiles)"""
        import numpy as np
        key = (cin, cout, in_stride, out_stride)
        hit = self._tables.get(key)
        if hit is None:
            tb = [ctypes.c_int() for _ in range(5)]
            rc = lib().sis3d_ragged_tiling_k3t16(cin, cout, self.T16_BRICK, *[ctypes.byref(v) for v in tb])
            if rc != 0:
                raise Sis3dUnsupported("RaggedBatch: no balanced k3 tiling for %d -> %d" % (cin, cout))
            bx, by, bz, ng, _ = (v.value for v in tb)
            rdt, _ = _rag_dtypes()
            nbk = -(-self.ext // np.array([bx, by, bz]))
            blks = np.concatenate([[0], np.cumsum(nbk.prod(1) * ng)])
            d = np.zeros(self.n, dtype=rdt)
            d["X"], d["Y"], d["Z"] = self.ext[:, 0], self.ext[:, 1], self.ext[:, 2]
            d["nbx"], d["nby"], d["nbz"] = nbk[:, 0], nbk[:, 1], nbk[:, 2]
            d["block0"] = blks[:-1]
            d["in_off"] = self.voffs[:-1] * in_stride
            d["out_off"] = self.voffs[:-1] * out_stride
            hit = (torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(self.device), int(blks[-1]))
            self._tables[key] = hit
        return hit

    def gather(self, vol):
        """crops of a channels-last volume (1,C,X,Y,Z) -> packed (voxels, C)"""
        C = vol.shape[1]
        out = torch.empty(self.voxels, C, device=vol.device)
        for i in range(self.n):
            x0, y0, z0, x1, y1, z1 = (int(v) for v in self.w[i])
            out[int(self.voffs[i]):int(self.voffs[i + 1])] = vol[0, :, x0:x1, y0:y1, z0:z1].permute(1, 2, 3, 0).reshape(-1, C)
        return out

    def conv(self, src, pc, relu=False, sigmoid=False):
        """one launch of a k1 / k3 conv over the whole batch: (voxels, cin) -> (voxels, cout).  k3: the balanced kernel's ragged
        launch; k1: a 1x1x1 conv does not see the crop structure -- the packed buffer is one (voxels x 1 x 1) activation."""
        cin_stride = src.shape[1]
        if cin_stride != pc.cin:
            raise _lib.Sis3dError("RaggedBatch.conv: activation has %d channels, packed weight expects %d" % (cin_stride, pc.cin))
        if pc.k == 1:
            x = src.view(1, self.voxels, 1, 1, cin_stride).permute(0, 4, 1, 2, 3)       # logical (1,C,V,1,1), channels-last memory
            y = conv3d(x, pc, relu=relu, sigmoid=sigmoid)
            return y.permute(0, 2, 3, 4, 1).reshape(self.voxels, pc.cout)
        if pc.packed_t16 is None or sigmoid:
            raise Sis3dUnsupported("RaggedBatch.conv: k3 layers need the balanced-kernel pack (cin % 32 == 0)")
        out = torch.empty(self.voxels, pc.cout, device=src.device)
        desc, blocks = self.table(pc.cin, pc.cout, cin_stride, pc.cout)
        check(lib().sis3d_conv3d_k3t16_ragged(_ptr(src), pc.cin, cin_stride, _ptr(pc.packed_t16), _ptr(pc.bias), pc.cout,
                                              EPI_RELU if relu else 0, _ptr(out), pc.cout, _ptr(desc), self.n, blocks, self.T16_BRICK, _stream()),
              "sis3d_conv3d_k3t16_ragged")
        return out

    def views(self, out):
        res = []
        NC = out.shape[1]
        for i in range(self.n):
            dx, dy, dz = (int(v) for v in self.ext[i])
            res.append(out[int(self.voffs[i]):int(self.voffs[i + 1])].view(dx, dy, dz, NC).permute(3, 0, 1, 2).unsqueeze(0))
        return res


def mask_head_batched(scene, windows, w0, pcs, pc_last, sigmoid=True):
    """The MaskBackbone (lib/nets/backbones.py:236-287) on ALL detected boxes at once: one launch per layer over a
    ragged batch of crops (sis3d_conv3d_planar2_ragged + 4 x sis3d_conv3d_ragged k3 + 1 x k1).
    scene (1,2,X,Y,Z) planar; windows [(x0,y0,z0,x1,y1,z1)]; w0 = conv0 weight (64,2,3,3,3); pcs = 4 PackedConv (64->64 k3);
    pc_last = PackedConv (64->NC, k1).  Returns a list of logical (1,NC,dx,dy,dz) tensors (views of one buffer)."""
    if len(windows) == 0:
        return []
    plan = MaskPlan(windows, pcs[0].cout, pc_last.cout, _dev(scene, "scene").device)
    return mask_head_run(scene, plan, w0, pcs, pc_last, sigmoid)


def maxpool3(x, out=None, out_coff=0):
    """nn.MaxPool3d(3,1,1); `out` / `out_coff`: write into a channel range of a wider channels-last tensor"""
    if not is_cl(x):
        raise _lib.Sis3dError("maxpool3 expects a channels-last activation")
    _, C, X, Y, Z = x.shape
    if out is None:
        out, out_coff = new_act(C, (X, Y, Z), x.device), 0
    elif not is_cl(out) or tuple(out.shape[2:]) != (X, Y, Z) or out_coff + C > out.shape[1]:
        raise _lib.Sis3dError("maxpool3: bad `out`")
    check(lib().sis3d_maxpool3d_3x3x3(_ptr(x), X, Y, Z, C, _ptr(out), out.shape[1], int(out_coff), _stream()), "sis3d_maxpool3d_3x3x3")
    return out
