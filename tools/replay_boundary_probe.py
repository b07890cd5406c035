#!/usr/bin/env python
"""What does the boundary between two replays of a pipeline's graph cost?  (rocprofv3 traces show each queue idle for ~100 us per
replay between the last kernel of one replay and the first of the next.)  Four pipelines as in bench.py; each pipeline's graph holds R
back-to-back passes over its chunk instead of one: same kernels, same streams, 1/R of the graph launches.
Usage: GPU_MAX_HW_QUEUES=8 python tools/replay_boundary_probe.py"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sis3d import ops, synthetic  # noqa: E402
from sis3d.engine import PipelinedEngines  # noqa: E402
from launch_probe import build_net, VOX  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    net = build_net()
    pe = PipelinedEngines(net, n, stage="rpn")
    for i in range(n):
        pe.load(i, synthetic.synth_chunk(i))
    pe.prepare(warmup=2)
    for R in (1, 2, 4, 8, 1):
        graphs = []
        ops.lib().sis3d_conv3d_k3t16_set_brick_cap(pe._brick_cap)
        ops.lib().sis3d_conv3d_k3wino_set_shared_chip(1)
        try:
            for eng, s in zip(pe.engines, pe.streams):
                with torch.cuda.stream(s), torch.no_grad():
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        for _ in range(R):
                            eng._step()
                    graphs.append(g)
            torch.cuda.synchronize()
        finally:
            ops.lib().sis3d_conv3d_k3t16_set_brick_cap(0)
            ops.lib().sis3d_conv3d_k3wino_set_shared_chip(0)

        def step():
            for g, s in zip(graphs, pe.streams):
                with torch.cuda.stream(s):
                    g.replay()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
        steps = max(25, 200 // R)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%d pipelines, %d passes per graph: %.4f ms per chunk, %.4g voxels/s (host in launches %.0f us per step)" %
              (n, R, dt / steps / (n * R) * 1e3, n * R * VOX * steps / dt, (t1 - t0) / steps * 1e6), flush=True)
        del graphs


if __name__ == "__main__":
    main()
