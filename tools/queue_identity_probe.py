"""Which of torch's pool streams share a hardware queue (with each other, with the null stream)?  A stream B that shares a queue
with stream A cannot start a kernel while A's long kernel runs (a hardware queue is in-order); on another queue it starts at once."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402


def main():
    torch.cuda.init()
    x = torch.zeros(64, device="cuda")
    null = torch.cuda.default_stream()
    pool = [torch.cuda.Stream() for _ in range(32)]
    names = ["null"] + ["p%d" % i for i in range(32)]
    streams = [null] + pool
    # calibrate the spin: ~2 ms
    cyc = 1_000_000
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cyc = int(cyc * 2e-3 / dt)
    t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("spin of %d cycles = %.2f ms" % (cyc, dt * 1e3))

    def blocked(a, b):
        """does a tiny kernel on b wait for a long kernel on a?"""
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

    classes = []          # list of lists of indices into streams
    for i, s in enumerate(streams):
        for cl in classes:
            if blocked(streams[cl[0]], s):
                cl.append(i)
                break
        else:
            classes.append([i])
    print("%d hardware-queue classes:" % len(classes))
    for k, cl in enumerate(classes):
        print("  queue class %d: %s" % (k, " ".join(names[i] for i in cl)))
    # a second pass the other way round for the first few (symmetry check)
    bad = 0
    for cl in classes:
        for i in cl[1:3]:
            if not blocked(streams[i], streams[cl[0]]):
                bad += 1
    print("asymmetric pairs:", bad)


if __name__ == "__main__":
    main()
