#!/bin/bash
set -u
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd "$ROOT" && timeout 900 python -m pytest tests/test_gpu_conv_t16.py tests/test_gpu_conv.py -x -q > "$OUT/pytest_conv.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_conv.log" )
tail -15 "$OUT/pytest_conv.log"
timeout 600 python "$ROOT/tools/t16_tune.py" rpn rpn_x2 g2_0 g2_b g1_b1 g1_b2 mask64 > "$OUT/t16_tune.log" 2>&1
cat "$OUT/t16_tune.log"
timeout 500 python "$ROOT/bench.py" --no-cpu-baseline 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_backbone_rpn.json"
cut -c1-400 "$OUT/bench_backbone_rpn.json"; tail -3 "$OUT/bench_default.err"
