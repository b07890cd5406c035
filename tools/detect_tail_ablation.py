#!/usr/bin/env python
"""What the detection tail costs the chip per chunk with four chunks in flight: four detect pipelines whose classifier stage is cut short
(no RoI pooling and no classifier / RoI pooling only / everything), beside four backbone + RPN pipelines.
Usage: python tools/detect_tail_ablation.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sis3d import ops, synthetic  # noqa: E402
from sis3d.engine import PipelinedEngines  # noqa: E402
from cu_time_ablation import timed, build_net  # noqa: E402


def main():
    n = 4
    net = build_net()
    full = type(net)._classify_rois
    shapes = {}

    def cut(mode):
        def f(self, l1, l2):
            if mode == "full" or "out" not in shapes:
                out = full(self, l1, l2)
                shapes.setdefault("out", [torch.zeros_like(t) for t in out])
                return out
            if mode == "pool":
                p = self._prop
                self._pool5 = ops.roi_pool_levels(l1, l2, p["rois"], p["levels"], self.cfg.CLASS_POOLING_SIZE, 1.0 / self._feat_stride[0],
                                                  out_channels_last=True)
            return tuple(shapes["out"])
        return f

    res = {}
    for label, stage, mode in (("backbone + RPN", "rpn", None), ("+ proposal layer (decode, top-k, NMS) + record packing", "detect", "none"),
                               ("+ RoI pooling", "detect", "pool"), ("+ classifier (fc1 split-K + tail) = detect", "detect", "full")):
        if mode is not None:
            type(net)._classify_rois = cut("full")
            pe0 = PipelinedEngines(net, 1, stage="detect")
            pe0.load(0, synthetic.synth_chunk(0))
            pe0.prepare(warmup=1)                       # one full pass: output shapes of the classifier
            del pe0
            type(net)._classify_rois = cut(mode)
        pe = PipelinedEngines(net, n, stage=stage)
        for i in range(n):
            pe.load(i, synthetic.synth_chunk(i))
        pe.prepare(warmup=2)
        res[label] = timed(pe.run, n, "four pipelines: " + label)
        del pe
    type(net)._classify_rois = full


if __name__ == "__main__":
    main()
