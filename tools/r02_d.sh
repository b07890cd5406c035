#!/bin/bash
set -u
TAG=${1:-r02d}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mv in 0 216 108 54; do
  for nf in 3 2; do
  SIS3D_K3_MAXVOX=$mv timeout 300 python "$ROOT/bench.py" --no-cpu-baseline --inflight $nf --steps 100 2>> "$OUT/err.log" | tail -1 > "$OUT/bench_maxvox${mv}_nf$nf.json"
  done
done
SIS3D_K3_LEGACY=1 timeout 300 python "$ROOT/bench.py" --no-cpu-baseline --steps 100 2>> "$OUT/err.log" | tail -1 > "$OUT/bench_legacy_nf3.json"
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
tail -5 "$OUT/err.log"
