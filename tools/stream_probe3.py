"""One process, one box: the streamed-input orders through the product API, then the scene three times (is it warm-up or mapping?)."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from sis3d import synthetic  # noqa: E402
from sis3d.engine import PipelinedEngines, _STREAM_POOL  # noqa: E402


def timed(step, steps=100, warm=20):
    for k in range(warm):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(warm + k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return "%.3f ms per step (host enqueue %.3f)" % ((t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3)


def main():
    net, cfg, sd = bench.build_net("backbone_rpn")
    n = 4
    eng = PipelinedEngines(net, n, stage="rpn")
    for i in range(n):
        eng.load(i, synthetic.synth_chunk(i))
    eng.prepare(warmup=2)
    bench.preheat(eng.run, 250.0)
    R = 4
    ring = [[synthetic.synth_chunk(100 + i * R + r).contiguous().pin_memory() for r in range(R)] for i in range(n)]
    print("resident                ", timed(lambda k: eng.run()))
    for copy in ("own", "per_pipeline"):
        eng.enable_feed("grid", copy=copy)
        for i in range(n):
            eng.feed(i, ring[i][0])

        def run_then_feed(k):
            for i in range(n):
                eng.run_fed(i)
                eng.feed(i, ring[i][(k + 1) % R])
        print("%-12s run->feed  " % copy, timed(run_then_feed))
        for i in range(n):
            eng.run_fed(i)
        torch.cuda.synchronize()

        def feed_then_run(k):
            for i in range(n):
                eng.run_fed(i, ring[i][k % R])
        print("%-12s feed->run  " % copy, timed(feed_then_run))

        def feed_all_then_run_all(k):
            for i in range(n):
                eng.feed(i, ring[i][k % R])
            for i in range(n):
                eng.run_fed(i)
        print("%-12s feedall/runall" % copy, timed(feed_all_then_run_all))
        torch.cuda.synchronize()
    # host cost of one H2D enqueue right behind a graph launch on the same stream vs on an idle stream
    st = eng.streams[0]
    buf = eng.engines[0].scenes[0]
    torch.cuda.synchronize()
    ts = []
    for rep in range(10):
        with torch.cuda.stream(st):
            eng.engines[0].run()
            t0 = time.perf_counter()
            buf.copy_(ring[0][rep % R], non_blocking=True)
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("H2D enqueue right behind a graph launch: host %.1f us (median)" % (sorted(ts)[5] * 1e6))
    ts = []
    for rep in range(10):
        with torch.cuda.stream(st):
            t0 = time.perf_counter()
            buf.copy_(ring[0][rep % R], non_blocking=True)
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("H2D enqueue on an idle stream:           host %.1f us (median)" % (sorted(ts)[5] * 1e6))
    del eng
    # ---- the scene, three times in a row, then with a long preheat
    import argparse
    args = bench.parse(["--steps", "20", "--warmup", "5"])
    dnet, dcfg, _ = bench.build_net("scene")
    sync = torch.cuda.synchronize
    r = None
    for rep in range(3):
        sc = bench.run_scene(dnet, args, 0, 1, 32, sync, steps=20, runner=r)
        r = sc["runner"]
        print("scene rep %d: %.3f ms per scene" % (rep, sc["dt"] / sc["steps"] * 1e3))
    args.preheat_ms = 4000.0
    sc = bench.run_scene(dnet, args, 0, 1, 32, sync, steps=20, runner=r, want_table=True)
    print("scene after a 250 ms preheat: %.3f ms per scene" % (sc["dt"] / sc["steps"] * 1e3))
    ss = bench.run_scene(dnet, args, 0, 1, 32, sync, steps=20, runner=r, streamed=True)
    print("scene streamed: %.3f ms per scene" % (ss["dt"] / ss["steps"] * 1e3))
    args.preheat_ms = 250.0
    for rep in range(2):
        sh = bench.run_scene(dnet, args, 0, 1, 32, sync, group="solo", steps=20, emulate=(0, 8), gathered=sc["table"])
        print("share rep %d: %.3f ms" % (rep, sh["dt"] / sh["steps"] * 1e3))
    print("streams:", {k: hex(v.cuda_stream) for k, v in _STREAM_POOL.items()})


if __name__ == "__main__":
    main()
