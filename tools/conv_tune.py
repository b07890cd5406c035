#!/usr/bin/env python
"""Micro-benchmark of sis3d.ops.conv3d on the network's layer shapes (HIP events on the launch stream).
Usage (GPU box): python tools/conv_tune.py [layer ...]; env SIS3D_K3_VARIANT selects tiling variants."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import ops  # noqa: E402

LAYERS = {  # name: (cin, cout, k, dims)
    "rpn": (128, 256, 3, (24, 12, 24)), "g2_0": (128, 128, 3, (24, 12, 24)), "g2_b": (64, 64, 3, (24, 12, 24)),
    "g1_b1": (32, 32, 3, (48, 24, 48)), "g1_b2": (32, 32, 3, (24, 12, 24)),
    "k1_32_32": (32, 32, 1, (48, 24, 48)), "k1_128_32": (128, 32, 1, (24, 12, 24)), "k1_32_128": (32, 128, 1, (24, 12, 24)),
    "k1_128_64": (128, 64, 1, (24, 12, 24)), "k1_64_128": (64, 128, 1, (24, 12, 24)), "head_88": (256, 88, 1, (24, 12, 24)),
    "rpn_512wg": (128, 256, 3, (32, 16, 16)), "rpn_256wg": (128, 256, 3, (16, 16, 16)), "rpn_1024wg": (128, 256, 3, (32, 32, 16)),
    "rpn_2048wg": (128, 256, 3, (32, 32, 32)), "rpn_768wg": (128, 256, 3, (32, 24, 16)),
    "k2_32_128": (32, 128, 2, (48, 24, 48)), "color0": (128, 64, 2, (96, 48, 96)),
}


def bench_batched():
    dims = (24, 12, 24)
    xs = [ops.new_act(128, dims, torch.device("cuda")).normal_() for _ in range(2)]
    pcs = [ops.PackedConv(torch.randn(256, 128, 3, 3, 3, device="cuda") * 0.05, torch.zeros(256, device="cuda")) for _ in range(2)]
    for _ in range(3):
        ops.conv3d_batched(xs, pcs, relu=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.conv3d_batched(xs, pcs, relu=True)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("rpn_x2 batched  %8.1f us  %6.1f TF" % (us, 2 * 2.0 * 6912 * 256 * 128 * 27 / us / 1e6))


def bench_bottlenecks():
    """the fused launches of the backbone: two Bottlenecks as FusedSequential (conv1 | conv2+conv3+res+next conv1 | ...)"""
    from sis3d.nets import backbones as bb
    for tag, planes_in, planes, dims in (("g1 32/32 @48x24x48", 32, 32, (48, 24, 48)), ("g1 128/32 @24x12x24", 128, 32, (24, 12, 24)),
                                         ("g2 128/64 @24x12x24", 128, 64, (24, 12, 24))):
        seq = bb.FusedSequential(bb.Bottleneck(planes_in, planes), bb.Bottleneck(planes_in, planes)).cuda().eval()
        x = ops.new_act(planes_in, dims, torch.device("cuda")).normal_()
        with torch.no_grad():
            for _ in range(3):
                seq(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    seq(x)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        print("bneck x2 %-22s big=%s small=%s  %8.1f us per pair" % (tag, os.environ.get("SIS3D_K3BIG_VARIANT", "0"),
                                                                  os.environ.get("SIS3D_K3_VARIANT", "0"), e0.elapsed_time(e1) / 20 * 1e3))


def main():
    names = sys.argv[1:] or list(LAYERS)
    if "bneck" in names:
        names.remove("bneck")
        bench_bottlenecks()
    if "rpn_x2" in names:
        names.remove("rpn_x2")
        bench_batched()
    for n in names:
        cin, cout, k, dims = LAYERS[n]
        x = ops.new_act(cin, dims, torch.device("cuda"))
        x.normal_()
        w = torch.randn(cout, cin, k, k, k, device="cuda") * 0.05
        pc = ops.PackedConv(w, torch.zeros(cout, device="cuda"))
        st = 2 if k == 2 else 1
        for _ in range(5):
            ops.conv3d(x, pc, stride=st, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 50
        # replay from a HIP graph so small kernels are not host-launch-bound
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.conv3d(x, pc, stride=st, relu=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(it):
                ops.conv3d(x, pc, stride=st, relu=True)
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
        od = [d // st for d in dims]
        fl = 2.0 * od[0] * od[1] * od[2] * cout * cin * k ** 3
        by = 4.0 * (dims[0] * dims[1] * dims[2] * cin + od[0] * od[1] * od[2] * cout + cout * cin * k ** 3)
        print("%-10s variant=%s  %8.1f us  %6.1f TF  %6.0f GB/s" % (n, os.environ.get("SIS3D_K3_VARIANT", "0") + "/" + os.environ.get("SIS3D_K1_VARIANT", "0"), us, fl / us / 1e6, by / us / 1e3))


if __name__ == "__main__":
    main()
