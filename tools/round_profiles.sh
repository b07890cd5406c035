#!/bin/bash
# Round-end measurements on the GPU box (run through gpurun): bench lines + rocprofv3 kernel stats -> gpurun_out/
# usage: tools/round_profiles.sh <tag>      (e.g. r01b)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { timeout 400 "$@"; }
run python "$ROOT/bench.py" 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_backbone_rpn.json"
run python "$ROOT/bench.py" --inflight 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_backbone_rpn_inflight1.json"
run python "$ROOT/bench.py" --workload detect --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_detect.json"
run python "$ROOT/bench.py" --workload images --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_images.json"
run python "$ROOT/bench.py" --workload images --from-depth --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_images_from_depth.json"
SIS3D_FORCE_DIST=1 run python "$ROOT/bench.py" --workload scene --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_scene.json"
for wl in backbone_rpn detect images; do
  rm -rf /tmp/prof_$wl
  run rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python "$ROOT/bench.py" --workload $wl --steps 100 --warmup 10 --no-cpu-baseline > /tmp/prof_$wl.log 2>&1
  f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/bench_${wl}_kernel_stats.csv"
  t=$(find /tmp/prof_$wl -name "*kernel_trace.csv" | head -1)
  # the stats row of the dominant kernel's template mixes three layers (216 / 432 / 864 workgroups); split the trace by grid
  [ -n "$t" ] && [ "$wl" = backbone_rpn ] && python "$ROOT/tools/dominant_from_trace.py" "$t" > "$OUT/dominant_kernel_from_trace.json"
done
for f in "$OUT"/bench_*.json; do echo "$(basename $f): $(cut -c1-160 $f)"; done
