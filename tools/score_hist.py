#!/usr/bin/env python
"""Distribution of the RPN candidate scores the bench's detect workload feeds to sis3d_topk_desc (why its fast path overflows)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
import bench  # noqa: E402
from sis3d import synthetic  # noqa: E402
from sis3d.engine import ChunkEngine  # noqa: E402

from sis3d import ops  # noqa: E402
net, cfg, sd = bench.build_net("detect")
CAP = {}
_real = ops.topk_desc


def _spy(scores, k):
    CAP["s"] = scores.detach().clone()
    return _real(scores, k)


ops.topk_desc = _spy
for cid in (0, 1, 5):
    eng = ChunkEngine(net, stage="detect", use_graph=False)
    eng.load(synthetic.synth_chunk(cid))
    eng.prepare(warmup=1)
    out = eng.run()
    torch.cuda.synchronize()
    s = CAP.get("s")
    if s is None:
        print("topk_desc was not called; keys:", list(out.keys()) if isinstance(out, dict) else type(out))
        break
    s = s.float().cpu()
    u, c = torch.unique(s, return_counts=True)
    srt = torch.sort(s, descending=True).values
    lo, hi = float(s.min()), float(s.max())
    b = ((s - lo) * (2047.0 / (hi - lo))).clamp(0, 2047).long()
    hist = torch.bincount(b, minlength=2048)
    cum = torch.flip(torch.cumsum(torch.flip(hist, [0]), 0), [0])
    print("chunk %d: n %d  max %.9g x%d  2nd %.9g x%d  400th %.9g  unique %d  >=0.999: %d  top-bucket %d  cum>=bucket(400th): %d" % (
        cid, s.numel(), float(u[-1]), int(c[-1]), float(u[-2]), int(c[-2]), float(srt[399]), u.numel(), int((s >= 0.999).sum()),
        int(hist[2047]), int(cum[int(b[torch.argsort(s, descending=True)[399]])])))
