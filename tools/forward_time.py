#!/usr/bin/env python
"""time the reference-shaped net.forward(blobs,'TEST',[]) (eager, with its host sync + mask branch) on the GPU box"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import config, synthetic
from sis3d.nets import backbones
for use_mask in (False, True):
    cfg = config.scannet_benchmark_cfg(); cfg.USE_MASK = use_mask
    net = backbones.ScanNet_Backbone(cfg=cfg); net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS)); net.cuda().eval()
    data = synthetic.synth_chunk(0)
    blobs = {"data": data, "id": ["x"], "gt_box": [torch.zeros(0, 7)], "gt_mask": [[]]}
    for _ in range(3): p = net.forward(blobs, "TEST", [])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): p = net.forward(blobs, "TEST", [])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    nm = len(p.get("mask_pred", [[]])[0]) if use_mask else 0
    vox = sum(int(m.shape[2] * m.shape[3] * m.shape[4]) for m in p["mask_pred"][0]) if use_mask else 0
    print("forward use_mask=%s: %.2f ms  rois=%d masks=%d mask_voxels=%d" % (use_mask, dt * 1e3, p["rois"][0].shape[0], nm, vox))
