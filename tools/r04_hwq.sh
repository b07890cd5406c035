#!/bin/bash
# more hardware queues (GPU_MAX_HW_QUEUES, default 4): does a 4th..6th chunk in flight stop losing?
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value %.4g' % d['value'], 'ms/chunk %.4f' % (d['ms_per_step'] / d['config']['chunks_per_step_per_gpu']))"; }
for q in 6 8 12 16 24; do for n in 3 4 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --inflight $n --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages --no-live-pmc 2>/dev/null | line "hwq=$q backbone_rpn inflight=$n"
done; done
for q in 8 16; do for n in 4; do for wl in "detect" "detect --masks" "images" "images --rgb"; do
  GPU_MAX_HW_QUEUES=$q python bench.py --workload $wl --inflight $n --steps 100 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages --no-live-pmc 2>/dev/null | line "hwq=$q $wl inflight=$n"
done; done; done
