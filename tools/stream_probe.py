"""Where does the streamed-input path lose time?  Variants of one step of four backbone+RPN pipelines (HIP events / wall clock):
  resident      graph replays only
  d2d           + device copy staging -> static on the pipeline stream before every replay
  events        + the event handshake with a copy stream (no H2D)
  h2d_shared    + the H2D on ONE shared copy stream (= PipelinedEngines.feed / run_fed)
  h2d_own       H2D on the pipeline's OWN stream straight into the static buffer (no staging, no copy stream)
  h2d_perpipe   feed / run_fed with one copy stream per pipeline
plus the raw pinned H2D rate of a 3.54 MB buffer and the host time of a step's enqueue."""
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
from sis3d.engine import PipelinedEngines  # noqa: E402


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
    return (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3


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
    stage = [[torch.empty_like(eng.engines[i].scenes[0]) for _ in range(2)] for i in range(n)]
    out = {}
    # raw H2D rate
    cs = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(cs):
        for _ in range(5):
            stage[0][0].copy_(ring[0][0], non_blocking=True)
        e0.record()
        for k in range(50):
            stage[0][k & 1].copy_(ring[0][k % R], non_blocking=True)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    out["raw_h2d_3.54MB"] = "%.1f us per copy = %.1f GB/s" % (ms * 1e3, 3.538944e-3 / ms)
    t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        for k in range(50):
            stage[0][k & 1].copy_(ring[0][k % R], non_blocking=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    out["raw_h2d_host_enqueue_us"] = (t1 - t0) / 50 * 1e6

    out["resident"] = timed(lambda k: eng.run())

    def d2d(k):
        for i in range(n):
            with torch.cuda.stream(eng.streams[i]):
                eng.engines[i].scenes[0].copy_(stage[i][k & 1], non_blocking=True)
                eng.engines[i].run()
    out["d2d"] = timed(d2d)

    ready = [[torch.cuda.Event() for _ in range(2)] for _ in range(n)]
    free = [[torch.cuda.Event() for _ in range(2)] for _ in range(n)]
    for i in range(n):
        for s in range(2):
            ready[i][s].record(cs)
            free[i][s].record(eng.streams[i])

    def events(k, h2d=False, streams=None):
        for i in range(n):
            s = k & 1
            st = eng.streams[i]
            c = streams[i] if streams else cs
            with torch.cuda.stream(st):
                st.wait_event(ready[i][s])
                eng.engines[i].scenes[0].copy_(stage[i][s], non_blocking=True)
                free[i][s].record(st)
                eng.engines[i].run()
            s2 = (k + 1) & 1
            c.wait_event(free[i][s2])
            with torch.cuda.stream(c):
                if h2d:
                    stage[i][s2].copy_(ring[i][(k + 1) % R], non_blocking=True)
                ready[i][s2].record(c)
    out["events"] = timed(events)
    out["h2d_shared"] = timed(lambda k: events(k, True))
    own = [torch.cuda.Stream() for _ in range(n)]
    for i in range(n):
        for s in range(2):
            ready[i][s].record(own[i])
    torch.cuda.synchronize()
    out["h2d_perpipe"] = timed(lambda k: events(k, True, own))

    def h2d_own(k):
        for i in range(n):
            with torch.cuda.stream(eng.streams[i]):
                eng.engines[i].scenes[0].copy_(ring[i][k % R], non_blocking=True)
                eng.engines[i].run()
    out["h2d_own"] = timed(h2d_own)
    # the product path
    eng.enable_feed("grid")
    for i in range(n):
        eng.feed(i, ring[i][0])

    def fed(k):
        for i in range(n):
            eng.run_fed(i)
            eng.feed(i, ring[i][(k + 1) % R])
    out["feed_run_fed"] = timed(fed)
    for k, v in out.items():
        print("%-24s %s" % (k, ("%.3f ms per step (host enqueue %.3f)" % v) if isinstance(v, tuple) else v))


if __name__ == "__main__":
    main()
