#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k maxpool 2>&1 | tail -3
for v in 0 1 0 1; do echo "SIS3D_POOL_LDS=$v"; SIS3D_POOL_LDS=$v python tools/pool_time.py 2>/dev/null | sed "s/zseg=auto/lds=$v/"; done
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'alone %.4f' % d['config']['single_chunk_latency_ms'])"; }
for rep in 1 2; do for v in 0 1; do
  SIS3D_POOL_LDS=$v python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages --no-live-pmc 2>/dev/null | line "pool_lds=$v"
done; done
