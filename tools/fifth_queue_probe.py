#!/usr/bin/env python
"""Does a FIFTH, light queue cost what a fifth heavy one does?  (five chunks in flight fall off a cliff on any GPU_MAX_HW_QUEUES:
profiles/r04_hw_queues.txt.)  Four pipelines of backbone + RPN as in bench.py, plus `k` small kernels (max-pool of a 24x12x24x128 map,
~5 us each) per step on a fifth stream, or on the null stream, or none.
Usage: GPU_MAX_HW_QUEUES=8 python tools/fifth_queue_probe.py"""
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
    n = 4
    net = build_net()
    pe = PipelinedEngines(net, n, stage="rpn")
    for i in range(n):
        pe.load(i, synthetic.synth_chunk(i))
    pe.prepare(warmup=2)
    x = ops.new_act(128, (24, 12, 24), torch.device("cuda")).normal_()
    y = ops.new_act(128, (24, 12, 24), torch.device("cuda"))
    side = torch.cuda.Stream()

    def small(k, stream):
        with torch.cuda.stream(stream):
            for _ in range(k):
                ops.maxpool3(x, y)

    for label, k, stream in (("4 pipelines only", 0, None), ("+ 8 small kernels / step on a 5th stream", 8, side),
                             ("+ 32 small kernels / step on a 5th stream", 32, side), ("+ 8 small kernels / step on pipeline 0's stream", 8, pe.streams[0]),
                             ("4 pipelines only (again)", 0, None)):
        def step():
            pe.run()
            if k:
                small(k, stream)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-52s %.4f ms/step  %.4g voxels/s" % (label, dt / 200 * 1e3, n * VOX * 200 / dt), flush=True)


if __name__ == "__main__":
    main()
