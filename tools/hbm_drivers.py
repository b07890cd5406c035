#!/usr/bin/env python
"""Eager drivers for the PMC passes of tools/hbm_pmc.sh: kernels that the detect bench does not launch.
  images : the materialised back-projection (proj_table / transpose / gather) + colour stem on the volume, 5 views
  misc   : tsdf_encode of a 96x48x96 chunk, compute_projection (frustum) of 5 views, planar mask-head first conv"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import config, ops, synthetic  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "misc"
    dev = torch.device("cuda")
    cfg = config.scannet_benchmark_cfg()
    dims = synthetic.CHUNK_DIMS
    if what == "images":
        feats, i3d, i2d = (t.to(dev) for t in synthetic.synth_views(0))
        for _ in range(20):
            vol = ops.project_views_max(feats, i3d, i2d, dims, ())
        x = vol
        pc = ops.PackedConv(torch.randn(64, 128, 2, 2, 2, device=dev) * 0.05, None)
        for _ in range(20):
            ops.conv3d(x, pc, stride=2, relu=True)
    else:
        sdf = (torch.randn(dims[0] * dims[1] * dims[2], device=dev) * 3).contiguous()
        for _ in range(20):
            ops.tsdf_encode(sdf, dims, 3.0, "abs", None, channels_last=False)
        from sis3d.layer_utils.projection import ProjectionHelper
        depth, c2w, w2g = synthetic.synth_cameras(1, 5, dims, cfg.VOXEL_SIZE)
        helper = ProjectionHelper(cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX, cfg.DEPTH_SHAPE, list(dims), cfg.VOXEL_SIZE)
        d = depth.to(dev)
        for _ in range(20):
            helper.compute_projection_views(d, c2w, w2g)
        x = ops.new_act(128, (24, 12, 24), dev).normal_()
        for _ in range(20):
            ops.maxpool3(x)
        x64 = ops.new_act(64, (48, 24, 48), dev).normal_()
        for _ in range(20):
            ops.maxpool3(x64)
        host = synthetic.synth_chunk(3).contiguous().pin_memory()
        dst = torch.empty(1, 2, *dims, device=dev)
        for _ in range(20):
            ops.upload(host, dst)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
