#!/usr/bin/env python
"""Time the mask head's first layer (sis3d_conv3d_planar2_ragged) alone: HIP events around 200 back-to-back launches over the crop batch of
bench.py --workload detect --masks (16 boxes); SIS3D_PLANAR_FMA=1 times the FMA kernel it replaced.
Usage: python tools/planar_mfma_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import ops, synthetic  # noqa: E402


def main():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    wins = []
    for _ in range(16):
        d = [int(torch.randint(8, 24, (1,), generator=g)) for _ in range(3)]
        d[1] = min(d[1], 16)
        o = [int(torch.randint(0, hi - dd + 1, (1,), generator=g)) for hi, dd in zip((96, 48, 96), d)]
        wins.append((o[0], o[1], o[2], o[0] + d[0], o[1] + d[1], o[2] + d[2]))
    plan = ops.MaskPlan(wins, 64, 19, dev)
    scene = synthetic.synth_chunk(0).cuda().float()
    w0 = torch.randn(64, 2, 3, 3, 3, device=dev) * 0.1
    st = scene.stride()

    def launch():
        ops.check(ops.lib().sis3d_conv3d_planar2_ragged(scene.data_ptr(), st[1], st[2], st[3], plan.gp.data_ptr(), plan.n, plan.items, w0.data_ptr(), 64,
                                                        1, plan.a.data_ptr(), 64, None), "planar2_ragged")
    for _ in range(20):
        launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            launch()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    print("SIS3D_PLANAR_FMA=%s: %d crops, %d voxels: %.2f us per launch" % (os.environ.get("SIS3D_PLANAR_FMA", "0"), plan.n, plan.voxels, best))


if __name__ == "__main__":
    main()
