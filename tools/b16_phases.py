#!/usr/bin/env python
"""In-kernel phase times of the split-bf16 k3 conv on the rpn_net layer (sis3d_conv3d_k3b16_set_trace: wall_clock64 of every wave at
the phase boundaries; 100 MHz ticks).  Usage (GPU box): python tools/b16_phases.py [brick]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import ops  # noqa: E402

brick = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = ops.lib()
x = ops.new_act(128, (24, 12, 24), "cuda")
x.normal_().clamp_(min=0)
w = torch.randn(256, 128, 3, 3, 3, device="cuda") * 0.05
pc = ops.PackedConv(w, torch.zeros(256, device="cuda"))
y = ops.new_act(256, (24, 12, 24), "cuda")
run = lambda: ops.conv3d_k3b16([x], [pc], relu=True, outs=[y], brick=brick)
for _ in range(10):
    run()
torch.cuda.synchronize()
nwg = 2048
buf = torch.zeros(nwg, 4, 16, dtype=torch.int64, device="cuda")
lib.sis3d_conv3d_k3b16_set_trace(ops._ptr(buf), nwg)
run()
torch.cuda.synchronize()
lib.sis3d_conv3d_k3b16_set_trace(None, 0)
t = buf.cpu().numpy()
used = t[:, 0, 0] > 0
t = t[used].astype(np.float64)
t0 = t[:, :, 0].min()
names = ["start", "prologue: loads, address math, convert + LDS stores", "barrier", "chunk 0 matrix loop", "barrier", "convert + store chunk 1, barrier",
         "chunk 1 matrix loop", "barrier", "store chunk 2, barrier", "chunk 2 matrix loop", "barrier", "store chunk 3, barrier",
         "chunk 3 matrix loop", "barrier", "(no further chunk)", "reduction + epilogue"]
print("brick %d: %d workgroups; kernel window %.1f us" % (brick, t.shape[0], (t[:, :, 15].max() - t0) / 100.0))
prev = t[:, :, 0]
for k in range(1, 16):
    cur = t[:, :, k]
    if (cur <= 0).all():
        continue
    d = (cur - prev) / 100.0
    print("  %-52s median %.2f us   (min %.2f  max %.2f)" % (names[k], np.median(d), d.min(), d.max()))
    prev = cur
