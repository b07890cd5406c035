"""time ops.maxpool3 on the network's three map shapes (graph replay of 50 launches); SIS3D_POOL_ZSEG forces a variant"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-sis_amd"))
from sis3d import ops
for C, dims in ((128, (24, 12, 24)), (64, (24, 12, 24)), (64, (48, 24, 48))):
    x = ops.new_act(C, dims, torch.device("cuda")).normal_()
    for _ in range(3): ops.maxpool3(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): ops.maxpool3(x)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    mb = 2 * C * dims[0] * dims[1] * dims[2] * 4 / 1e6
    print("zseg=%s C=%d dims=%s  %.1f us  %.0f GB/s" % (os.environ.get("SIS3D_POOL_ZSEG", "auto"), C, dims, us, mb / us * 1e3))
