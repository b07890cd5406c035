#!/bin/bash
# experiment batch 1: (a) geometry2[0] forced to two cout tiles per workgroup (108 work items) with 1 / 3 chunks in flight,
# (b) queue gaps of the 3-in-flight run
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r04_exp1}; mkdir -p "$OUT"
cd "$ROOT"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'alone %.4f' % d['config']['single_chunk_latency_ms'])"; }
for rep in 1 2; do
for nc in 0 2; do for n in 1 3; do
  SIS3D_WINO_NC=$nc python bench.py --inflight $n --steps 200 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages 2>/dev/null | line "nc=$nc inflight=$n"
done; done; done | tee "$OUT/nc_sweep.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profg
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/profg -- python "$ROOT/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages > /tmp/profg.log 2>&1
t=$(find /tmp/profg -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/trace_gaps.py" "$t" | tee "$OUT/gaps_inflight3.txt"
head -3 "$t" > "$OUT/trace_head.csv"
