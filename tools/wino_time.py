#!/usr/bin/env python
"""Winograd F(2x2x2,3x3x3) fp32 kernel (sis3d_conv3d_k3wino) against the direct fp32 MFMA kernel (sis3d_conv3d_k3t16) on the
network's k3 layer shapes: HIP-graph replay of 20 launches, HIP events, best of 3.  Usage (GPU box): python tools/wino_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sis3d import ops  # noqa: E402
from t16_tune import LAYERS, timeit  # noqa: E402


def main():
    dev = torch.device("cuda")
    names = sys.argv[1:] or ["rpn", "rpn_x2", "g2_0", "g2_b", "g1_b1", "g1_b2", "mask64"]
    for n in names:
        key, nprob = (n[:-3], 2) if n.endswith("_x2") else (n, 1)
        cin, cout, dims = LAYERS[key]
        xs = [ops.new_act(cin, dims, dev).normal_().clamp_(min=0) for _ in range(nprob)]
        pcs = [ops.PackedConv(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05, torch.zeros(cout, device=dev)) for _ in range(nprob)]
        fl = nprob * 2.0 * dims[0] * dims[1] * dims[2] * cout * cin * 27
        ops.set_winograd(False)
        t_d = timeit(lambda: ops.conv3d_k3t16(xs, pcs, relu=True))
        ops.set_winograd(True)
        t_w = timeit(lambda: ops.conv3d_k3wino(xs, pcs, relu=True))
        a = ops.conv3d_k3wino(xs, pcs, relu=True)[0]
        ops.set_winograd(False)
        d = ops.conv3d_k3t16(xs, pcs, relu=True)[0]
        ops.set_winograd(True)
        err = float((a - d).abs().max())
        print("%-8s x%d  direct %7.1f us (%5.1f TF)   winograd %7.1f us (%5.1f TF algorithmic, %5.1f TF executed = %2.0f %% of 157.3)   "
              "max |diff| %.1e" % (key, nprob, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6, fl / 3.375 / t_w / 1e6,
                                   100 * fl / 3.375 / t_w / 1e6 / 157.3, err), flush=True)


if __name__ == "__main__":
    main()
