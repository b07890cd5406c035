#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value %.4g' % d['value'], 'us/chunk %.2f' % (1e3 * d['ms_per_step'] / d['config']['chunks_per_step_per_gpu']), 'alone %.4f' % d['config']['single_chunk_latency_ms'])"; }
for rep in 1 2; do for t in 0 2 4 8; do
  SIS3D_PW_TPW=$t python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages --no-live-pmc 2>/dev/null | line "tpw=$t"
done; done
