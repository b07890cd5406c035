#!/bin/bash
# shared-chip dispatch incl. the Bottleneck(128, 32) bodies on the Winograd kernel: tests, then the headline A/B (SIS3D_BNECK_WINO=0 keeps the
# direct bodies everywhere -- incl. the 48x24x48 ones -- so the A/B of the 24x12x24 bodies alone is the pair of SHARED_MIN runs)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r04_exp3}; mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_bottleneck.py tests/test_gpu_conv_wino.py -x -q -m gpu 2>&1 | tail -5 | tee "$OUT/pytest.txt"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'alone %.4f' % d['config']['single_chunk_latency_ms'])"; }
for rep in 1 2; do
for wl in backbone_rpn detect; do
  python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages 2>/dev/null | line "default $wl"
done; done | tee "$OUT/headline.txt"
for n in 2 4 5 6; do
  python bench.py --inflight $n --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages 2>/dev/null | line "inflight=$n"
done | tee -a "$OUT/headline.txt"
