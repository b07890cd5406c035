#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_enet.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python bench.py --workload images --rgb --steps 100 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages --no-live-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rgb value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'alone %.4f' % d['config']['single_chunk_latency_ms'], 'enet_ms_5_views %.4f' % d['config']['enet_ms_5_views'])"; done
