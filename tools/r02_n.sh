#!/bin/bash
set -u
TAG=${1:-r02n}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { timeout 500 "$@"; }
run python "$ROOT/bench.py" 2> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_backbone_rpn.json"
run python "$ROOT/bench.py" --inflight 1 --no-cpu-baseline 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_backbone_rpn_inflight1.json"
run python "$ROOT/bench.py" --workload detect --no-cpu-baseline 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_detect.json"
run python "$ROOT/bench.py" --workload detect --masks --no-cpu-baseline 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_detect_masks.json"
run python "$ROOT/bench.py" --workload images --no-cpu-baseline 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_images.json"
run python "$ROOT/bench.py" --workload images --rgb --no-cpu-baseline 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_images_rgb.json"
run python "$ROOT/bench.py" --workload images --from-depth --no-cpu-baseline 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_images_from_depth.json"
SIS3D_FORCE_DIST=1 run python "$ROOT/bench.py" --workload scene --steps 20 --warmup 10 --no-cpu-baseline 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_scene.json"
SIS3D_FORCE_DIST=1 run python "$ROOT/bench.py" --workload scene --scene-chunks 4 --steps 100 --warmup 20 --no-cpu-baseline 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_scene4.json"
python - "$OUT" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f.split("/")[-1], "unreadable", e); continue
    st = d.get("stages", {})
    print("%-36s value %7.1f M  ms/step %.3f  single %.3f  backbone %.3f rpn %.3f  dom %.1f us %s" % (
        f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], d["config"].get("single_chunk_latency_ms") or 0,
        st.get("backbone", {}).get("ms", 0), st.get("rpn", {}).get("ms", 0), d["roofline"]["launch_us"],
        {k: v for k, v in d["config"].items() if k in ("enet_graph_captured", "mask_head_gflop", "records_gathered")}))
PY
tail -3 "$OUT/err.log"
