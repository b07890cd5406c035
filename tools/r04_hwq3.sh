#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value %.4g' % d['value'], 'ms/chunk %.4f' % (d['ms_per_step'] / d['config']['chunks_per_step_per_gpu']))"; }
for n in 4 5 6 8; do for wl in "images --rgb" "detect --masks"; do
  GPU_MAX_HW_QUEUES=12 python bench.py --workload $wl --inflight $n --steps 100 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages --no-live-pmc 2>/dev/null | line "hwq=12 $wl inflight=$n"
done; done
