#!/bin/bash
# in-flight throughput vs k3 brick cap and chunks in flight (default bench workload)
for cap in 0 108 216; do for nf in 2 3 4; do
  v=$(SIS3D_K3_MAXVOX=$cap timeout 200 python bench.py --inflight $nf --steps 150 --warmup 20 --no-cpu-baseline --no-stages 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M vox/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))")
  echo "cap $cap inflight $nf: $v"
done; done
