"""Not a test (not collected): times the device compute_projection (5 views, chunk + whole-scene grids) beside the torch-CPU
oracle.  Lives under tests/ because it executes the oracle, which only tests / smoke() / bench's cpu_baseline may do."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from sis3d import config, synthetic, ops
from sis3d.layer_utils.projection import ProjectionHelper
import sis3d_oracle as orc
c = config.scannet_benchmark_cfg()
for dims in ((96, 48, 96), (256, 96, 320)):
    h = ProjectionHelper(c.INTRINSIC, c.PROJ_DEPTH_MIN, c.PROJ_DEPTH_MAX, c.DEPTH_SHAPE, list(dims), c.VOXEL_SIZE)
    depth, c2w, w2g = synthetic.synth_cameras(1, 5, dims, c.VOXEL_SIZE)
    d = depth.cuda()
    params = torch.stack([h.view_params(c2w[v], w2g[v]) for v in range(5)]).cuda()
    nvox = dims[0] * dims[1] * dims[2]
    out = (torch.empty(5, nvox + 1, dtype=torch.int64, device="cuda"), torch.empty(5, nvox + 1, dtype=torch.int64, device="cuda"))
    run = lambda: ops.compute_projection(d, params, dims, c.DEPTH_SHAPE, c.INTRINSIC, c.PROJ_DEPTH_MIN, c.PROJ_DEPTH_MAX, c.VOXEL_SIZE, out=out)
    for _ in range(5): run()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    gpu_ms = e0.elapsed_time(e1) / 50
    t = time.perf_counter(); full = h.compute_projection_views(d, c2w, w2g); torch.cuda.synchronize(); host_ms = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    for v in range(5):
        orc.compute_projection(depth[v], c2w[v], w2g[v], c.INTRINSIC, c.PROJ_DEPTH_MIN, c.PROJ_DEPTH_MAX, c.DEPTH_SHAPE, dims, c.VOXEL_SIZE)
    cpu_ms = (time.perf_counter() - t) * 1e3
    wr = 2 * 5 * (nvox + 1) * 8
    print("dims %s  5 views: kernels %.3f ms (%.0f GB/s of list writes)  incl. host geometry %.2f ms  torch-CPU oracle %.1f ms  counts %s"
          % (dims, gpu_ms, wr / gpu_ms / 1e6, host_ms, cpu_ms, full[0][:, 0].tolist()), flush=True)
