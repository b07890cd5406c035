#!/bin/bash
set -u
TAG=${1:-r02i}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd "$ROOT" && timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" )
tail -4 "$OUT/pytest.log"
timeout 500 python "$ROOT/bench.py" --no-cpu-baseline --workload detect 2>> "$OUT/err.log" | tail -1 > "$OUT/bench_detect_nf3.json"
python - "$OUT" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    d = json.load(open(f))
    print("%-28s value %7.1f M  ms/step %.3f  single %.3f" % (f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], d["config"]["single_chunk_latency_ms"] or 0))
PY
rm -rf /tmp/prof_b
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python "$ROOT/bench.py" --workload detect --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-stages > /tmp/prof_b.log 2>&1
t=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/detect_inflight1_by_grid.md"
grep -E "topk|mlp_tail|roi_pool|fc_splitk|nms_|decode|Fill|pack_records" "$OUT/detect_inflight1_by_grid.md" | cut -c1-60,100-170
