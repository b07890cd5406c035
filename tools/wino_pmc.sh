#!/bin/bash
# PMC passes (counters only with --kernel-trace) over the Winograd k3 kernel: clock, MFMA busy, wait breakdown, LDS, HBM traffic
set -u
TAG=${1:-r03wp}
LAYER=${2:-rpn}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rm -rf /tmp/wpmc_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/wpmc_$i -- python "$ROOT/tools/wino_pmc.py" $LAYER > /tmp/wpmc_$i.log 2>&1
  f=$(find /tmp/wpmc_$i -name "*counter_collection.csv" | head -1)
  t=$(find /tmp/wpmc_$i -name "*kernel_trace.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/pass_${i}_counters.csv"; [ -n "$t" ] && cp "$t" "$OUT/pass_${i}_trace.csv"; echo "pass $i [$ctrs] ok"; else echo "pass $i: no counter file"; tail -5 /tmp/wpmc_$i.log; fi
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
summary = {}
for f in sorted(glob.glob(sys.argv[1] + "/pass_*_counters.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if "k3wino" in r.get("Kernel_Name", ""):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    tr = f.replace("_counters", "_trace")
    dur = []
    try:
        for r in csv.DictReader(open(tr)):
            if "k3wino" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    except Exception:
        pass
    dur.sort()
    md = dur[len(dur) // 2] if dur else float("nan")
    print(f.split("/")[-1], "median duration us", md)
    for k, v in agg.items():
        v = sorted(v)
        print("   ", k, "launches", len(v), "median", v[len(v) // 2])
        summary[k] = {"median": v[len(v) // 2], "launches": len(v), "median_duration_us_in_this_pass": md}
    if "GRBM_GUI_ACTIVE" in agg and dur:
        g = sorted(agg["GRBM_GUI_ACTIVE"])[len(agg["GRBM_GUI_ACTIVE"]) // 2]
        print("    effective clock GHz", g / md / 1e3)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    summary["traffic_bytes_per_launch"] = int(summary["FETCH_SIZE"]["median"] * 1024 * 2 + summary["WRITE_SIZE"]["median"] * 1024)
    summary["fetch_correction"] = "x2 (gfx950: FETCH_SIZE counts 64 B per 128 B request on wide coalesced reads, MI355X_MICROARCH.md HBM section)"
json.dump(summary, open(sys.argv[1] + "/pmc_summary.json", "w"), indent=1)
PY
