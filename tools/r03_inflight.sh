#!/bin/bash
# inflight sweep after the round-3 Winograd kernel work: value (G voxels/s) per workload and number of chunks in flight
run() { python bench.py --steps 100 --no-side-workloads --no-cpu-baseline "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']/1e9,4), 'G voxels/s', round(d['ms_per_step'],4), 'ms')"; }
for n in 3 4 5; do run --workload backbone_rpn --inflight $n; done
for n in 3 4; do run --workload detect --inflight $n; done
for n in 3 4; do run --workload detect --masks --inflight $n; done
for n in 3 4; do run --workload images --inflight $n; done
for n in 3 4; do run --workload scene --inflight $n --steps 20; done
