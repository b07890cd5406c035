#!/usr/bin/env python
"""Driver for PMC passes over the balanced k3 kernel: N eager launches of one layer (rocprofv3 wraps this script).
Usage: python tools/t16_pmc.py <layer> [nprob] [brick] [zero]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sis3d import ops  # noqa: E402
from t16_tune import LAYERS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "rpn"
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 1
brick = int(sys.argv[3]) if len(sys.argv) > 3 else -1
zero = len(sys.argv) > 4 and sys.argv[4] == "zero"
cin, cout, dims = LAYERS[name]
dev = torch.device("cuda")
xs = [ops.new_act(cin, dims, dev).normal_().clamp_(min=0) for _ in range(nprob)]
ws = [torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05 for _ in range(nprob)]
if zero:
    for x in xs:
        x.zero_()
    for w in ws:
        w.zero_()
pcs = [ops.PackedConv(w, torch.zeros(cout, device=dev)) for w in ws]
for _ in range(60):
    ops.conv3d_k3t16(xs, pcs, relu=True, brick=brick)
torch.cuda.synchronize()
