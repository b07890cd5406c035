#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r04_exp4}; mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_bottleneck.py tests/test_gpu_conv_wino.py tests/test_gpu_network.py tests/test_mask_variants.py -x -q -m gpu 2>&1 | tail -4 | tee "$OUT/pytest.txt"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'alone %.4f' % d['config']['single_chunk_latency_ms'], 'roofline %.4f %.2f us' % (d['roofline']['frac'], d['roofline']['launch_us']))"; }
for rep in 1 2; do
for wl in backbone_rpn detect; do
  python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages --no-live-pmc 2>/dev/null | line "default $wl"
done; done | tee "$OUT/headline.txt"
python bench.py --inflight 1 --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-live-pmc 2>/dev/null | line "inflight=1" | tee -a "$OUT/headline.txt"
