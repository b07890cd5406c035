#!/bin/bash
# shared-chip rule of the Winograd dispatch (csrc/conv3d_wino.hip wino_nc): work items a layer needs to take the kernel with two cout
# tiles per workgroup when several chunks are in flight.  1000 = rule off (r3 behaviour), 100 = geometry2[0] only, 48 = + the 64 -> 64
# convs of geometry2's Bottlenecks (54 work items each)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r04_exp2}; mkdir -p "$OUT"
cd "$ROOT"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'alone %.4f' % d['config']['single_chunk_latency_ms'])"; }
for rep in 1 2; do
for m in 1000 100 48; do for wl in backbone_rpn detect; do
  SIS3D_WINO_SHARED_MIN=$m python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages 2>/dev/null | line "shared_min=$m $wl"
done; done; done | tee "$OUT/shared_min.txt"
for m in 1000 48; do for n in 2 4 6; do
  SIS3D_WINO_SHARED_MIN=$m python bench.py --inflight $n --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages 2>/dev/null | line "shared_min=$m inflight=$n"
done; done | tee -a "$OUT/shared_min.txt"
