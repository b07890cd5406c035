#!/usr/bin/env python
"""Driver for PMC passes over the Winograd k3 kernel: N eager launches of one layer (rocprofv3 wraps this script).
Usage: python tools/wino_pmc.py <layer> [nprob]"""
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
cin, cout, dims = LAYERS[name]
dev = torch.device("cuda")
xs = [ops.new_act(cin, dims, dev).normal_().clamp_(min=0) for _ in range(nprob)]
pcs = [ops.PackedConv(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05, torch.zeros(cout, device=dev)) for _ in range(nprob)]
for _ in range(40):
    ops.conv3d_k3wino(xs, pcs, relu=True)
torch.cuda.synchronize()
