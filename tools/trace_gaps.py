#!/usr/bin/env python
"""rocprofv3 kernel_trace.csv -> how busy each HIP queue was: per queue, the busy time (sum of kernel durations), the idle time
BETWEEN consecutive kernels of that queue, and the distribution of those gaps; plus the union over queues (time at least one kernel
was running).  With several chunks in flight the gaps are what a stream spends waiting for the command processor / for CUs.
Usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n_rows]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"][:60]) for r in rows), key=lambda e: e[0])
    ev = ev[skip:]
    perq = collections.defaultdict(list)
    for s, e, q, n in ev:
        perq[q].append((s, e, n))
    print("queues:", len(perq))
    for q, lst in sorted(perq.items(), key=lambda kv: -len(kv[1])):
        if len(lst) < 50:
            continue
        busy = sum(e - s for s, e, _ in lst)
        gaps = [max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)]
        gaps_small = sorted(g for g in gaps if g < 200000)            # drop the host-side pauses between timed regions
        span = lst[-1][1] - lst[0][0]
        n = len(gaps_small)
        pct = lambda p: gaps_small[min(n - 1, int(p * n))] / 1e3 if n else 0.0
        print("queue %s: %d kernels, span %.1f ms, busy %.1f ms (%.0f %%), gaps<200us: n %d sum %.1f ms median %.2f us p90 %.2f us p99 %.2f us" %
              (q, len(lst), span / 1e6, busy / 1e6, 100.0 * busy / span, n, sum(gaps_small) / 1e6, pct(0.5), pct(0.9), pct(0.99)))
        # gap that FOLLOWS each kernel type (mean), top 8
        after = collections.defaultdict(list)
        for i, g in enumerate(gaps):
            if g < 200000:
                after[lst[i][2]].append(g / 1e3)
        for name, v in sorted(after.items(), key=lambda kv: -sum(kv[1]))[:8]:
            print("    after %-60s n %5d mean gap %.2f us" % (name, len(v), sum(v) / len(v)))
    # union busy
    tot, cur_s, cur_e = 0, None, None
    for s, e, _, _ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    print("union of all kernels: %.1f ms of %.1f ms" % (tot / 1e6, (ev[-1][1] - ev[0][0]) / 1e6))


if __name__ == "__main__":
    main()
