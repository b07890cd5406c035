#!/usr/bin/env python
"""time ops.nms_select / topk / classifier under graph replay (GPU box)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import ops

def timeit(fn, it=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

g = torch.Generator().manual_seed(0)
m = 33394
lo = torch.rand(m, 3, generator=g) * torch.tensor([80.0, 40.0, 80.0])
boxes = torch.cat([lo, lo + torch.rand(m, 3, generator=g) * 20 + 1], 1).cuda()
scores = torch.rand(m, generator=g).cuda()
lv = torch.ones(m).cuda()
s_sorted, order = ops.topk_desc(scores, 400)
print("nms_select n=400: %.1f us" % timeit(lambda: ops.nms_select(boxes, lv, s_sorted, order, 400, 0.1, 200)))
print("topk 33394->400: %.1f us" % timeit(lambda: ops.topk_desc(scores, 400)))
print("torch.sort 33394: %.1f us" % timeit(lambda: torch.sort(scores, descending=True, stable=True)))
