"""streams created by US (hipStreamCreateWithPriority through ctypes, wrapped as torch.cuda.ExternalStream) before
torch's pool is touched -- do four pipelines on them land on four distinct hardware queues every time, and how do they compare with
the best window of torch's pool that calibrate() finds?   python tools/own_streams_probe.py [n_streams]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def own_streams(n, priority=None):
    hip = ctypes.CDLL("libamdhip64.so")
    lo, hi = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    out = []
    for _ in range(n):
        s = ctypes.c_void_p()
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(s), ctypes.c_uint(1), ctypes.c_int(lo.value if priority is None else priority))   # 1 = hipStreamNonBlocking
        assert rc == 0, rc
        out.append(torch.cuda.ExternalStream(s.value))
    return out, (lo.value, hi.value)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    mine, prange = own_streams(n)
    print("GPU_MAX_HW_QUEUES=%s  priority range %s  %d own streams created before torch's pool" % (os.environ.get("GPU_MAX_HW_QUEUES"), prange, n))
    x = torch.zeros(64, device="cuda")
    cyc = 1_000_000
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cyc = int(cyc * 2e-3 / dt)
    t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); dt = time.perf_counter() - t0

    def blocked(a, b):
        torch.cuda.synchronize()
        with torch.cuda.stream(a):
            torch.cuda._sleep(cyc)
        with torch.cuda.stream(b):
            t0 = time.perf_counter()
            x.add_(1.0)
            b.synchronize()
            d = time.perf_counter() - t0
        torch.cuda.synchronize()
        return d > 0.5 * dt
    streams = [torch.cuda.default_stream()] + mine
    names = ["null"] + ["own%d" % i for i in range(n)]
    classes = []
    for i, s in enumerate(streams):
        for cl in classes:
            if blocked(streams[cl[0]], s):
                cl.append(i)
                break
        else:
            classes.append([i])
    print("%d hardware-queue classes: %s" % (len(classes), " | ".join(" ".join(names[i] for i in cl) for cl in classes)))
    # four pipelines of backbone + RPN on the own streams vs the calibrated window of torch's pool
    import bench
    from sis3d import synthetic
    from sis3d.engine import PipelinedEngines
    net, cfg, _ = bench.build_net("backbone_rpn")
    pe = PipelinedEngines(net, 4, stage="rpn")
    for i in range(4):
        pe.load(i, synthetic.synth_chunk(i))
    pe.prepare(warmup=2, calibrate=False)

    def timed(label, steps=200):
        for _ in range(20):
            pe.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pe.run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print("%-46s %.4f ms per step of 4 chunks" % (label, ms), flush=True)
        return ms
    timed("torch pool, first window (no calibration)")
    for w in range(0, n - 3):
        pe.streams = mine[w:w + 4]
        timed("own streams %d..%d" % (w, w + 3))
    best, times = pe.calibrate(pe.run, reps=3, warm=1)
    print("calibrate(): window %d of torch's pool, %s" % (best, {k: round(v, 4) for k, v in times.items()}))
    timed("torch pool, calibrated window %d" % best)


if __name__ == "__main__":
    main()
