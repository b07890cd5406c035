#!/usr/bin/env python
"""Mask head alone (BASELINE config 3's largest FLOP consumer): the ragged one-launch-per-layer batch on the bench's
detection set (the first 16 non-degenerate RoIs of synthetic chunk 0), graph-replayed, for every k3t16 brick.
Usage (GPU box): python tools/mask_time.py [n_boxes]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sis3d import ops, synthetic  # noqa: E402
from sis3d.engine import ChunkEngine  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    net, cfg, sd = bench.build_net("detect", masks=True)
    eng = ChunkEngine(net, stage="detect", use_graph=False, mask_boxes=nb)
    data = synthetic.synth_chunk(0)
    eng.load(data)
    eng.prepare(warmup=1)
    eng.run()
    torch.cuda.synchronize()
    plan0 = eng.mask_plan
    print("boxes", plan0.n, "voxels", plan0.voxels, "GFLOP %.2f" % (plan0.flops / 1e9), "dims", plan0.dims[:6])
    scene = data.cuda().float()
    mb = net.mask_backbone
    for brick in ((2, 3, 4, 5) if os.environ.get("SIS3D_T16_NOCLIP") else (-1, 1, 2, 3, 4, 5)):
        if brick >= 0:
            os.environ["SIS3D_MASK_BRICK"] = str(brick)
        else:
            os.environ.pop("SIS3D_MASK_BRICK", None)
        w = [tuple(int(v) for v in x) for x in plan0.windows] if hasattr(plan0, "windows") else None
        if w is None:
            print("MaskPlan keeps no windows; add them"); return
        plan = mb.plan(w, scene.device)
        for _ in range(3):
            mb.forward_planned(scene, plan)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            mb.forward_planned(scene, plan)
        for _ in range(5):
            g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 50
        k3 = 2.0 * plan.voxels * 4 * 27 * 64 * 64
        import numpy as np
        bd = {1: (6, 6, 6), 2: (3, 6, 6), 3: (3, 3, 6), 4: (4, 4, 4), 5: (4, 4, 8)}[plan.brick_t16]
        mt = -(-(bd[0] * bd[1] * bd[2]) // 16)
        grp = 3 if mt % 3 == 0 else (4 if mt >= 4 else mt)
        full = run = 0
        for e in plan.dims:
            ax = [np.minimum(bd[k], e[k] - bd[k] * np.arange(-(-e[k] // bd[k]))) for k in range(3)]
            vox = ax[0][:, None, None] * ax[1][None, :, None] * ax[2][None, None, :]
            full += vox.size * mt
            run += int(np.minimum(mt, -(-(-(-vox // 16)) // grp) * grp).sum())
        print("   tiles: %d fixed-brick, %d with clipping (%.0f %%), ideal %d" % (full, run, 100.0 * run / full, -(-plan.voxels // 16)))
        print("brick %2d (picked %d): %.3f ms  %.1f TF total, k3 layers' share of FLOPs %.0f %%, blocks %d" % (
            brick, plan.brick_t16, ms, plan.flops / ms / 1e9, 100 * k3 / plan.flops, plan.blocks_t16))


if __name__ == "__main__":
    main()
