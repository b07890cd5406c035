#!/bin/bash
# Round 3, first GPU call: the GPU test suite with the [parity] lines kept, the default bench line (both workloads), the N = 2 line
# on one shared GPU (functional: gloo instead of RCCL).
set -u
TAG=${1:-r03a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
rm -f "$OUT/parity_log.txt"
SIS3D_PARITY_LOG="$OUT/parity_log.txt" timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -15 "$OUT/pytest_gpu.log"
echo "---- parity log"; cat "$OUT/parity_log.txt"
timeout 400 python bench.py --steps 20 --warmup 5 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_default.json"
echo "---- bench default"; cut -c1-3000 "$OUT/bench_default.json"; tail -5 "$OUT/bench_default.err"
SIS3D_BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2> "$OUT/bench_n2_shared.err" | tail -1 > "$OUT/bench_n2_shared.json"
echo "---- bench N=2 shared GPU (functional)"; cut -c1-2500 "$OUT/bench_n2_shared.json"; tail -5 "$OUT/bench_n2_shared.err"
