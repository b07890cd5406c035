"""Mirror of lib/layer_utils/generate_anchors.py:58-119 (host numpy, cached per shape)."""
import numpy as np

from ..config import anchor_sizes, cfg as _default_cfg

_cache = {}


def anchors_for_level(size, stride, sizes):
    """anchor[k*A + a] = (-s_a/2, +s_a/2) + stride*(i,j,k); K voxels in 'ij' order (z fastest)."""
    key = (tuple(int(v) for v in size), int(stride), tuple(tuple(float(x) for x in s) for s in sizes))
    if key not in _cache:
        base = np.zeros((len(sizes), 6))
        for i, s in enumerate(sizes):
            base[i, 0:3] = [-float(s[0]) / 2, -float(s[1]) / 2, -float(s[2]) / 2]
            base[i, 3:6] = [float(s[0]) / 2, float(s[1]) / 2, float(s[2]) / 2]
        gx, gy, gz = np.meshgrid(np.arange(0, size[0]) * stride, np.arange(0, size[1]) * stride,
                                 np.arange(0, size[2]) * stride, indexing="ij")
        shifts = np.stack([gx.ravel(), gy.ravel(), gz.ravel()] * 2, axis=1)
        a = (base[None, :, :] + shifts[:, None, :]).reshape(-1, 6).astype(np.float32)
        a.setflags(write=False)                            # shared by every caller of this shape
        _cache[key] = a
    return _cache[key]


def generate_anchors(size_level1, size_level2, size_level3, feat_stride, cfg=None):
    cfg = cfg or _default_cfg
    out = []
    for lv, size in ((1, size_level1), (2, size_level2), (3, size_level3)):
        if cfg["NUM_ANCHORS_LEVEL%d" % lv] != 0:
            out.append(anchors_for_level(size, feat_stride[lv - 1], anchor_sizes(cfg, lv)))
        else:
            out.append(None)
    return tuple(out)
