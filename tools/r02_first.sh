#!/bin/bash
# Round-2 first GPU call: GPU tests, the new bench lines (stages / scene / masks), un-overlapped kernel trace.
set -u
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { timeout 500 "$@"; }
( cd "$ROOT" && timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log" )
tail -5 "$OUT/pytest_gpu.log"
run python "$ROOT/bench.py" 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_backbone_rpn.json"
run python "$ROOT/bench.py" --workload detect --no-cpu-baseline 2>"$OUT/bench_detect.err" | tail -1 > "$OUT/bench_detect.json"
run python "$ROOT/bench.py" --workload detect --masks --no-cpu-baseline 2>"$OUT/bench_detect_masks.err" | tail -1 > "$OUT/bench_detect_masks.json"
SIS3D_FORCE_DIST=1 run python "$ROOT/bench.py" --workload scene --steps 20 --warmup 10 --no-cpu-baseline 2>"$OUT/bench_scene.err" | tail -1 > "$OUT/bench_scene.json"
for nf in 3 4 2; do
SIS3D_FORCE_DIST=1 run python "$ROOT/bench.py" --workload scene --scene-chunks 4 --inflight $nf --steps 100 --warmup 20 --no-cpu-baseline 2>>"$OUT/bench_scene4.err" | tail -1 > "$OUT/bench_scene4_inflight$nf.json"
done
for wl in backbone_rpn detect; do
  rm -rf /tmp/prof_$wl
  run rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python "$ROOT/bench.py" --workload $wl --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-stages > /tmp/prof_$wl.log 2>&1
  f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/bench_${wl}_inflight1_kernel_stats.csv"
  t=$(find /tmp/prof_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/bench_${wl}_inflight1_by_grid.md"
done
for f in "$OUT"/bench_*.json; do echo "$(basename $f): $(cut -c1-200 $f)"; done
