#!/usr/bin/env python
"""Where the CU-time of a chunk goes when four chunks are in flight (the regime of the headline): per-chunk time of four pipelines that
run (a) the backbone proper, (b) backbone + RPN (the headline); and of single kernels replayed on four streams at once -- what a layer costs
the chip per chunk when everything around it is also busy.
Usage: GPU_MAX_HW_QUEUES=8 python tools/cu_time_ablation.py"""
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
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def build_net():
    return bench.build_net("detect")[0]


def timed(step, chunks, label, steps=200):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        for _ in range(4):
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-64s %8.2f us per chunk" % (label, dt / steps / chunks * 1e6), flush=True)
    return dt / steps / chunks * 1e6


def four_streams(fn, n=4):
    """capture fn() on each of n streams -> step() replaying all of them"""
    streams = [torch.cuda.Stream() for _ in range(n)]
    graphs = []
    # r5 ABI: the dispatch regime is a per-call argument of the library / a thread-local value on the Python side
    with ops.dispatch_regime(shared_chip=True, brick_cap=108):
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.no_grad():
                fn()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    fn()
                graphs.append(g)
        torch.cuda.synchronize()

    def step():
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
    return step, graphs


def main():
    n = 4
    net = build_net()
    res = {}
    for stage in ("backbone", "rpn"):
        pe = PipelinedEngines(net, n, stage=stage)
        for i in range(n):
            pe.load(i, synthetic.synth_chunk(i))
        pe.prepare(warmup=2)
        res[stage] = timed(pe.run, n, "four pipelines, stage %s" % stage)
        del pe
    print("   -> RPN (conv pair + heads) = %.2f us per chunk" % (res["rpn"] - res["backbone"]))
    dev = torch.device("cuda")

    def conv(cin, cout, dims, nprob=1):
        xs = [ops.new_act(cin, dims, dev).normal_().clamp_(min=0) for _ in range(nprob)]
        pcs = [ops.PackedConv(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05, torch.zeros(cout, device=dev)) for _ in range(nprob)]
        return lambda: ops.conv3d_k3t16(xs, pcs, relu=True)
    for label, fn in (("rpn conv pair (2 x 128->256 @24x12x24, Winograd, 432 work items)", conv(128, 256, (24, 12, 24), 2)),
                      ("geometry2[0] (128->128 @24x12x24, shared-chip form: 108 work items)", conv(128, 128, (24, 12, 24))),
                      ("64->64 @24x12x24 (shared-chip form: 54 Winograd work items)", conv(64, 64, (24, 12, 24)))):
        step, keep = four_streams(fn)
        timed(step, n, "x4 streams: " + label)
        del keep
    x = ops.new_act(128, (24, 12, 24), dev).normal_()
    step, keep = four_streams(lambda: ops.maxpool3(x))
    timed(step, n, "x4 streams: max-pool 24x12x24x128")


if __name__ == "__main__":
    main()
