#!/usr/bin/env python
"""Driver for PMC passes over the split-bf16 k3 kernel: 60 eager launches of the rpn_net layer (rocprofv3 wraps this script)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import ops  # noqa: E402

brick = int(sys.argv[1]) if len(sys.argv) > 1 else -1
x = ops.new_act(128, (24, 12, 24), "cuda")
x.normal_().clamp_(min=0)
w = torch.randn(256, 128, 3, 3, 3, device="cuda") * 0.05
pc = ops.PackedConv(w, torch.zeros(256, device="cuda"))
y = ops.new_act(256, (24, 12, 24), "cuda")
for _ in range(60):
    ops.conv3d_k3b16([x], [pc], relu=True, outs=[y], brick=brick)
torch.cuda.synchronize()
