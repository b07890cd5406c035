#!/usr/bin/env python
"""Is the chunk pipeline bound by the host's graph launches?  (the rocprofv3 trace of the 3-in-flight run shows ~90 us of idle queue
between the last kernel of one replay and the first of the next, tools/trace_gaps.py)
  A  the bench loop: n graphs (one per pipeline / stream), n hipGraphLaunch per step; host time in the launch calls vs total
  B  ONE graph per step: the capture stream forks into the n pipeline streams (optionally staggered), `rounds` passes each
Usage: python tools/launch_probe.py [inflight] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import config, ops, synthetic  # noqa: E402
from sis3d.engine import PipelinedEngines  # noqa: E402
from sis3d.nets import backbones  # noqa: E402

VOX = 96 * 48 * 96


def build_net():
    cfg = config.scannet_benchmark_cfg()
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS))
    return net.cuda().eval()


def timed(fn, steps, label, chunks):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        for _ in range(16):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s host-in-launch %.1f us/step, total %.1f us/step, %.4f ms/chunk, %.4g voxels/s" %
          (label, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6, (t2 - t0) / steps / chunks * 1e3, chunks * VOX * steps / (t2 - t0)), flush=True)


def round_graph(pe, rounds, stagger):
    main = torch.cuda.Stream()
    main.wait_stream(torch.cuda.current_stream())
    ops.lib().sis3d_conv3d_k3t16_set_brick_cap(pe._brick_cap)
    try:
        with torch.no_grad():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main):
                prev_ev = None
                for e, (eng, s) in enumerate(zip(pe.engines, pe.streams)):
                    s.wait_stream(main)
                    with torch.cuda.stream(s):
                        for r in range(rounds):
                            if prev_ev is not None and stagger and r == 0:
                                s.wait_event(prev_ev)
                            ev = torch.cuda.Event()
                            eng.net._after_level1 = ev.record if r == 0 else None
                            try:
                                eng._step()
                            finally:
                                eng.net._after_level1 = None
                            if r == 0:
                                prev_ev = ev
                for s in pe.streams:
                    main.wait_stream(s)
        torch.cuda.synchronize()
    finally:
        ops.lib().sis3d_conv3d_k3t16_set_brick_cap(0)
    return g, main


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    net = build_net()
    pe = PipelinedEngines(net, n, stage="rpn")
    for i in range(n):
        pe.load(i, synthetic.synth_chunk(i))
    pe.prepare(warmup=2)
    timed(pe.run, steps, "A: %d graphs / step (bench loop)" % n, n)
    # a single engine's replay, host cost alone
    e0, s0 = pe.engines[0], pe.streams[0]

    def one():
        with torch.cuda.stream(s0):
            e0.run()
    timed(one, steps, "A1: 1 graph / step", 1)
    for rounds, stagger in ((1, False), (1, True), (2, True), (4, True), (4, False)):
        g, main_s = round_graph(pe, rounds, stagger)

        def rep():
            with torch.cuda.stream(main_s):
                g.replay()
        timed(rep, max(20, steps // rounds), "B: one graph = %d pipelines x %d rounds%s" % (n, rounds, ", staggered" if stagger else ""), n * rounds)
        del g


if __name__ == "__main__":
    main()
