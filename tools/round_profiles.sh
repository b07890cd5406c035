#!/bin/bash
# Round-6 judged artefacts in ONE gpurun call (one profile refresh per round):
#   full GPU suite with the [parity] log, the driver-style bench line, rocprofv3 kernel stats + per-grid tables (four chunks in flight
#   and one chunk alone; detect and images-from-RGB alone), the dominant kernel from the trace, the N = 2 functional line on one GPU,
#   PMC passes of the dominant (Winograd) kernel and the HBM table of the memory-bound kernels.
#   usage: bash tools/round_profiles.sh <tag> [nosuite|tail]      -> gpurun_out/<tag>/   (copy what is judged into profiles/)
set -u
TAG=${1:-r06}
MODE=${2:-full}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
Q="--no-cpu-baseline --no-live-pmc --no-side-workloads --no-side-configs --no-streamed"
if [ "$MODE" = full ]; then
  rm -f "$OUT/parity_log.txt"
  SIS3D_PARITY_LOG="$OUT/parity_log.txt" timeout 900 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest_gpu.txt" 2>&1
  echo "pytest rc $?" >> "$OUT/parity_log.txt"
  tail -3 "$OUT/pytest_gpu.txt" >> "$OUT/parity_log.txt"
  tail -3 "$OUT/pytest_gpu.txt"
fi
timeout 500 python bench.py --steps 20 --warmup 5 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_driver_style.json"
# N = 2 ranks on this one GPU (gloo instead of RCCL): the N-rank code path end to end -- a functional line, not a measurement
[ "$MODE" != tail ] && SIS3D_BENCH_SHARE_GPU=1 timeout 500 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-live-pmc 2> "$OUT/bench_n2.err" | tail -1 > "$OUT/bench_n2_shared_gpu_functional.json"
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <name> <bench args...>: kernel trace + stats -> by-grid table
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python "$ROOT/bench.py" "$@" --steps 100 --warmup 10 --no-stages $Q > /tmp/prof_$name.log 2>&1
  grep "^{" /tmp/prof_$name.log | tail -1 > "$OUT/bench_${name}_under_rocprof.json"
  local f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/bench_${name}_kernel_stats.csv"
  local t=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/bench_${name}_by_grid.md"
  [ -n "$t" ] && [ "$name" = backbone_rpn ] && python "$ROOT/tools/dominant_from_trace.py" "$t" "$OUT/bench_${name}_under_rocprof.json" > "$OUT/dominant_kernel_from_trace.json"
}
if [ "$MODE" = tail ]; then      # only the tables of the detection tail and the mask head (kernels changed after the round's refresh)
  prof detect_inflight1 --workload detect --inflight 1
  prof detect --workload detect
  prof detect_masks_inflight1 --workload detect --masks --inflight 1
  cd "$ROOT"
  python tools/show_line.py "$OUT/bench_driver_style.json"
  exit 0
fi
prof backbone_rpn --workload backbone_rpn
prof backbone_rpn_inflight1 --workload backbone_rpn --inflight 1
prof detect_inflight1 --workload detect --inflight 1
prof images_rgb_inflight1 --workload images --rgb --inflight 1
prof detect --workload detect
prof detect_masks_inflight1 --workload detect --masks --inflight 1
cd "$ROOT"
# PMC: counters only with --kernel-trace, FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md)
bash tools/wino_pmc.sh "$TAG/wino_pmc" rpn > "$OUT/wino_pmc.log" 2>&1
bash tools/hbm_pmc.sh "$TAG/hbm" > "$OUT/hbm_pmc.log" 2>&1
python tools/show_line.py "$OUT/bench_driver_style.json"
tail -4 "$OUT/wino_pmc.log"
tail -30 "$OUT/hbm_pmc.log"
