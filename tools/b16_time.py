#!/usr/bin/env python
"""rpn_net k3 128->256 conv: exact-fp32 balanced kernel vs the split-bf16 variant (graph replay of 20 launches, best of 3).
Usage (GPU box): python tools/b16_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import ops  # noqa: E402

ops.lib()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20 * 1e3)
    return best


for cin, cout, dims, name in ((128, 256, (24, 12, 24), "rpn_net 128->256"), (128, 128, (24, 12, 24), "geometry2[0] 128->128"),
                              (64, 64, (24, 12, 24), "Bottleneck conv2 64->64"), (32, 32, (48, 24, 48), "Bottleneck conv2 32->32 @48")):
    x = ops.new_act(cin, dims, "cuda")
    x.normal_().clamp_(min=0)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.02
    pc = ops.PackedConv(w, torch.zeros(cout, device="cuda"))
    y = ops.new_act(cout, dims, "cuda")
    flop = 2.0 * dims[0] * dims[1] * dims[2] * cin * cout * 27
    t = timed(lambda: ops.conv3d_k3t16([x], [pc], relu=True, outs=[y]))
    print("%-24s fp32 t16            %7.1f us  %6.1f TF" % (name, t, flop / t / 1e6))
    for brick in (1, 2, 3, 4, 5):
        t = timed(lambda: ops.conv3d_k3b16([x], [pc], [w], relu=True, outs=[y], brick=brick))
        print("%-24s split-bf16 brick %d  %7.1f us  %6.1f TF-equivalent" % (name, brick, t, flop / t / 1e6))
