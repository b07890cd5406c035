#!/usr/bin/env python
"""Micro-benchmark of the balanced k3 kernel (sis3d_conv3d_k3t16) on the network's layer shapes: every brick, HIP-graph
replay of 20 launches, HIP events.  Usage (GPU box): python tools/t16_tune.py [layer ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import ops  # noqa: E402

LAYERS = {"rpn": (128, 256, (24, 12, 24)), "g2_0": (128, 128, (24, 12, 24)), "g2_b": (64, 64, (24, 12, 24)),
          "g1_b1": (32, 32, (48, 24, 48)), "g1_b2": (32, 32, (24, 12, 24)), "mask64": (64, 64, (30, 30, 36))}
BRICKS = {0: "6x6x12", 1: "6x6x6", 2: "3x6x6", 3: "3x3x6", 4: "4x4x4", 5: "4x4x8", 6: "4x8x8"}


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(it):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it * 1e3)
    return best


def main():
    names = sys.argv[1:] or list(LAYERS)
    dev = torch.device("cuda")
    for n in names:
        nprob = 1
        key = n
        if n.endswith("_x2"):
            key, nprob = n[:-3], 2
        cin, cout, dims = LAYERS[key]
        xs = [ops.new_act(cin, dims, dev).normal_().clamp_(min=0) for _ in range(nprob)]
        pcs = [ops.PackedConv(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05, torch.zeros(cout, device=dev)) for _ in range(nprob)]
        fl = nprob * 2.0 * dims[0] * dims[1] * dims[2] * cout * cin * 27
        auto = ops.lib().sis3d_conv3d_k3t16_brick(dims[0], dims[1], dims[2], cin, cout, nprob, ops.regime()[1])      # r5 ABI: max_voxels = the thread's brick cap
        for b in sorted(BRICKS):
            us = timeit(lambda: ops.conv3d_k3t16(xs, pcs, relu=True, brick=b))
            print("%-8s x%d brick %-7s %s %8.1f us  %6.1f TF  (%.0f %% of 157.3)" % (key, nprob, BRICKS[b], "*" if b == auto else " ", us,
                                                                                     fl / us / 1e6, fl / us / 1e6 / 1.573))
        # the 32x32-tile kernel of conv3d.hip on the same problem
        for pc in pcs:
            pc.t16_saved, pc.packed_t16 = pc.packed_t16, None
        us = timeit(lambda: ops.conv3d_batched(xs, pcs, relu=True) if nprob > 1 else ops.conv3d(xs[0], pcs[0], relu=True))
        print("%-8s x%d conv3d.hip (32x32 tiles)   %8.1f us  %6.1f TF" % (key, nprob, us, fl / us / 1e6))
    # Bottleneck pairs: split path (k3t16 + pointwise chain) vs the fused-chain launches
    from sis3d.nets import backbones as bb
    for tag, planes_in, planes, dims in (("g1 32/32 @48x24x48", 32, 32, (48, 24, 48)), ("g1 128/32 @24x12x24", 128, 32, (24, 12, 24)),
                                         ("g2 128/64 @24x12x24", 128, 64, (24, 12, 24))):
        seq = bb.FusedSequential(bb.Bottleneck(planes_in, planes), bb.Bottleneck(planes_in, planes)).cuda().eval()
        x = ops.new_act(planes_in, dims, dev).normal_()
        with torch.no_grad():
            for split in (True, False):
                bb.SPLIT_BOTTLENECK = split
                us = timeit(lambda: seq(x))
                print("bneck x2 %-22s %-6s %8.1f us per pair" % (tag, "split" if split else "fused", us))
        bb.SPLIT_BOTTLENECK = True
    # the pointwise half alone: register-chained (pointwise.hip) vs the generic 1x1x1 path with fused stage (conv3d.hip)
    import torch.nn.functional as F  # noqa: F401
    for tag, c0, c1, c2, dims in (("pw 32->32(+res)->32 @48", 32, 32, 32, (48, 24, 48)), ("pw 32->128(+res)->32 @24", 32, 128, 32, (24, 12, 24)),
                                  ("pw 64->128(+res)->64 @24", 64, 128, 64, (24, 12, 24)), ("pw 32->32 plain @48", 32, 32, None, (48, 24, 48)),
                                  ("pw 128->64 plain @24", 128, 64, None, (24, 12, 24))):
        y2 = ops.new_act(c0, dims, dev).normal_()
        res = ops.new_act(c1, dims, dev).normal_() if c2 is not None else None
        pc3 = ops.PackedConv(torch.randn(c1, c0, 1, 1, 1, device=dev) * 0.1, torch.zeros(c1, device=dev))
        stage = dict(pc=ops.PackedConv(torch.randn(c2, c1, 1, 1, 1, device=dev) * 0.1, torch.zeros(c2, device=dev)), relu=True) if c2 else None
        nv = dims[0] * dims[1] * dims[2]
        mb = 4e-6 * nv * (c0 + c1 * (2 if res is not None else 1) + (c2 or 0))
        us = timeit(lambda: ops.conv3d_pw16(y2, pc3, residual=res, relu=True, stage=stage))
        saved, pc3.packed_pw16 = pc3.packed_pw16, None
        us_old = timeit(lambda: ops.conv3d_pw_chain(y2, pc3, residual=res, relu=True, stage=stage) if c2 is not None
                        else ops.conv3d(y2, pc3, relu=True))
        pc3.packed_pw16 = saved
        print("%-28s pw16 %6.1f us (%4.0f GB/s of %.1f MB)   conv3d.hip %6.1f us" % (tag, us, mb / us * 1e3, mb, us_old))


if __name__ == "__main__":
    main()
