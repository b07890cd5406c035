#!/bin/bash
# Round-4 judged artefacts, one gpurun call: full GPU suite with the [parity] log, bench lines, rocprofv3 kernel stats / per-grid
# tables (all chunks in flight and one chunk alone), PMC passes of the dominant (Winograd) kernel.
#   usage: bash tools/r04_profiles.sh <tag> [quick]        -> gpurun_out/<tag>/
set -u
TAG=${1:-r04}
MODE=${2:-full}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() { timeout 500 "$@"; }
if [ "$MODE" = full ]; then
  rm -f "$OUT/parity_log.txt"
  SIS3D_PARITY_LOG="$OUT/parity_log.txt" timeout 900 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest_gpu.txt" 2>&1
  echo "pytest rc $?" >> "$OUT/parity_log.txt"
  tail -3 "$OUT/pytest_gpu.txt" >> "$OUT/parity_log.txt"
  tail -3 "$OUT/pytest_gpu.txt"
fi
run python bench.py --steps 20 --warmup 5 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_driver_style.json"
run python bench.py 2>> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_backbone_rpn.json"
run python bench.py --inflight 1 --no-cpu-baseline --no-side-workloads --no-split-line 2>/dev/null | tail -1 > "$OUT/bench_backbone_rpn_inflight1.json"
run python bench.py --workload detect --no-cpu-baseline --no-split-line 2>/dev/null | tail -1 > "$OUT/bench_detect.json"
if [ "$MODE" = full ]; then
  run python bench.py --workload detect --masks --no-cpu-baseline --no-split-line 2>/dev/null | tail -1 > "$OUT/bench_detect_masks.json"
  run python bench.py --workload images --no-cpu-baseline --no-split-line 2>/dev/null | tail -1 > "$OUT/bench_images.json"
  run python bench.py --workload images --rgb --no-cpu-baseline --no-split-line 2>/dev/null | tail -1 > "$OUT/bench_images_rgb.json"
fi
cd /tmp && export TMPDIR=/tmp
for wl in backbone_rpn detect; do
  if [ "$wl" = backbone_rpn ]; then
    rm -rf /tmp/prof_$wl
    timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python "$ROOT/bench.py" --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-side-workloads --no-split-line > /tmp/prof_$wl.log 2>&1
    grep "^{" /tmp/prof_$wl.log | tail -1 > "$OUT/bench_${wl}_under_rocprof.json"
    f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/bench_${wl}_kernel_stats.csv"
    t=$(find /tmp/prof_$wl -name "*kernel_trace.csv" | head -1)
    [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/bench_${wl}_by_grid.md"
    [ -n "$t" ] && python "$ROOT/tools/dominant_from_trace.py" "$t" "$OUT/bench_${wl}_under_rocprof.json" > "$OUT/dominant_kernel_from_trace.json"
  fi
  rm -rf /tmp/prof1_$wl
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$wl -- python "$ROOT/bench.py" --workload $wl --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-stages --no-side-workloads --no-split-line > /tmp/prof1_$wl.log 2>&1
  t=$(find /tmp/prof1_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/bench_${wl}_inflight1_by_grid.md"
done
cd "$ROOT"
bash tools/r03_wpmc.sh "$TAG/wino_pmc" rpn > "$OUT/wino_pmc.log" 2>&1
for f in "$OUT"/bench_*.json; do echo "$(basename $f): $(cut -c1-220 $f)"; done
tail -4 "$OUT/wino_pmc.log"
