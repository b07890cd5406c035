#!/bin/bash
# HBM traffic of the memory-bound kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel-trace only
# (MI355X_MICROARCH.md, HBM section), over the detect pass (one chunk in flight) and a few single-kernel drivers.
set -u
TAG=${1:-r02_hbm}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  for what in detect images images_rgb misc rpn; do
    rm -rf /tmp/hp_${ctr}_$what
    case $what in
      detect) cmd="python $ROOT/bench.py --workload detect --inflight 1 --steps 30 --warmup 5 --no-cpu-baseline --no-stages --no-graph --no-live-pmc --no-side-workloads";;
      images) cmd="python $ROOT/tools/hbm_drivers.py images";;
      images_rgb) cmd="python $ROOT/bench.py --workload images --rgb --inflight 1 --steps 12 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-live-pmc --no-side-workloads";;
      misc) cmd="python $ROOT/tools/hbm_drivers.py misc";;
      rpn) cmd="python $ROOT/tools/t16_pmc.py rpn 1";;
    esac
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/hp_${ctr}_$what -- $cmd > /tmp/hp_${ctr}_$what.log 2>&1
    f=$(find /tmp/hp_${ctr}_$what -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python "$ROOT/tools/hbm_table.py" --reduce "$f" > "$OUT/${what}_${ctr}.json"; echo "$what $ctr ok"; else echo "$what $ctr: no counters"; tail -3 /tmp/hp_${ctr}_$what.log; fi
  done
done
python "$ROOT/tools/hbm_table.py" --table "$OUT" > "$OUT/hbm_kernels.json"
python - "$OUT/hbm_kernels.json" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1]))["kernels"]:
    print("%-52s %7.1f us  algo %6.2f MB  pmc %6.2f MB (rd %6.2f wr %6.2f) ratio %5.2f  %5.0f GB/s algo = %4.1f %% of 8 TB/s" % (
        r["kernel"][:52], r["us"], r["algorithmic_mb"] or 0, r["pmc_mb"], r["fetch_mb"], r["write_mb"], r.get("traffic_ratio") or 0,
        r["algo_gbs"] or 0, 100 * (r["hbm_frac"] or 0)))
PY
