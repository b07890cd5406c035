#!/bin/bash
# Round-3 judged artefacts: bench lines of every workload, rocprofv3 kernel stats / per-grid tables of the default bench command,
# PMC passes of the dominant (Winograd) kernel, FETCH_SIZE / WRITE_SIZE table of the HBM-bound kernels.
set -u
TAG=${1:-r03_final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() { timeout 500 "$@"; }
run python bench.py 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_backbone_rpn.json"
run python bench.py --inflight 1 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 > "$OUT/bench_backbone_rpn_inflight1.json"
run python bench.py --workload detect --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_detect.json"
run python bench.py --workload detect --masks --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_detect_masks.json"
run python bench.py --workload images --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_images.json"
run python bench.py --workload images --rgb --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_images_rgb.json"
run python bench.py --inflight 3 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 > "$OUT/bench_backbone_rpn_inflight3.json"
SIS3D_FORCE_DIST=1 run python bench.py --workload scene --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_scene.json"
SIS3D_FORCE_DIST=1 run python bench.py --workload scene --scene-chunks 4 --steps 100 --warmup 20 --no-cpu-baseline --no-side-workloads 2>/dev/null | tail -1 > "$OUT/bench_scene4.json"
cd /tmp && export TMPDIR=/tmp
for wl in backbone_rpn detect; do
  rm -rf /tmp/prof_$wl
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python "$ROOT/bench.py" --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-side-workloads > /tmp/prof_$wl.log 2>&1
  grep "^{" /tmp/prof_$wl.log | tail -1 > "$OUT/bench_${wl}_under_rocprof.json"
  f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/bench_${wl}_kernel_stats.csv"
  t=$(find /tmp/prof_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/bench_${wl}_by_grid.md"
  if [ -n "$t" ] && [ "$wl" = backbone_rpn ]; then
    python "$ROOT/tools/dominant_from_trace.py" "$t" "$OUT/bench_${wl}_under_rocprof.json" > "$OUT/dominant_kernel_from_trace.json"
    python "$ROOT/tools/dominant_from_trace.py" --direct "$t" > "$OUT/direct_kernel_from_trace.json"
  fi
  rm -rf /tmp/prof1_$wl
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$wl -- python "$ROOT/bench.py" --workload $wl --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-stages --no-side-workloads --no-split-line > /tmp/prof1_$wl.log 2>&1
  t=$(find /tmp/prof1_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/bench_${wl}_inflight1_by_grid.md"
done
cd "$ROOT"
bash tools/r03_wpmc.sh "$TAG/wino_pmc" rpn > "$OUT/wino_pmc.log" 2>&1
# issue-cost microbenchmark and the Winograd kernel's phase timestamps / one-cost-at-a-time variants (tools/_bin: built on the build host by
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/issue_overlap.hip -o tools/_bin/issue_overlap
#   hipcc ... -fno-slp-vectorize -Iinclude -I3d-sis_amd/csrc -DWN_EXP=<bits> tools/wino_bench.cpp 3d-sis_amd/csrc/conv3d_wino.hip 3d-sis_amd/csrc/api.hip)
if [ -x tools/_bin/issue_overlap ]; then timeout 120 tools/_bin/issue_overlap > "$OUT/issue_overlap.txt" 2>&1; fi
: > "$OUT/wino_phases.txt"
for b in 0 96 119 103 118 117 115 112; do
  if [ -x tools/_bin/wino_bench_$b ]; then
    for a in "128 256 24 12 24 1" "128 256 24 12 24 2" "128 128 24 12 24 1"; do timeout 60 tools/_bin/wino_bench_$b $a | tail -2 >> "$OUT/wino_phases.txt"; done
  fi
done
bash tools/hbm_pmc.sh "$TAG/hbm" > "$OUT/hbm_pmc.log" 2>&1
for f in "$OUT"/bench_*.json; do echo "$(basename $f): $(cut -c1-200 $f)"; done
tail -12 "$OUT/hbm_pmc.log"
