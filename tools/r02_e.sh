#!/bin/bash
set -u
TAG=${1:-r02e}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd "$ROOT" && timeout 900 python -m pytest tests/test_gpu_conv_t16.py tests/test_gpu_conv.py tests/test_gpu_network.py -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" )
tail -12 "$OUT/pytest.log"
timeout 600 python "$ROOT/tools/t16_tune.py" rpn g1_b1 > "$OUT/t16_tune.log" 2>&1
grep -E "\*|bneck|pw " "$OUT/t16_tune.log"
for nf in 3 1; do
timeout 500 python "$ROOT/bench.py" --no-cpu-baseline --inflight $nf 2>> "$OUT/err.log" | tail -1 > "$OUT/bench_nf$nf.json"
done
SIS3D_K3_MAXVOX=108 timeout 500 python "$ROOT/bench.py" --no-cpu-baseline 2>> "$OUT/err.log" | tail -1 > "$OUT/bench_maxvox108_nf3.json"
python - "$OUT" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    st = d.get("stages", {})
    print("%-28s value %7.1f M  ms/step %.3f  single %.3f  stages: backbone %.3f rpn %.3f  dominant %.1f us" % (
        f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], d["config"]["single_chunk_latency_ms"] or 0,
        st.get("backbone", {}).get("ms", 0), st.get("rpn", {}).get("ms", 0), d["roofline"]["launch_us"]))
PY
