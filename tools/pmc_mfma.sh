#!/bin/bash
# MFMA-pipe utilisation of the dominant kernel from PMC counters (separate passes, kernel-trace only): run through gpurun
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_mfma
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$i -- python "$ROOT/tools/conv_tune.py" rpn > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/pass_${i}_$(echo $ctr | tr ' ' '_').csv"; echo "pass $i $ctr: $(wc -l < $f) rows"; else echo "pass $i $ctr: no counter file"; tail -3 /tmp/pmc_$i.log; fi
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/pass_*.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if "conv3d_mfma_kernel<3, 1, 4, 4, 4, 2, 2, 3, 1, 32, false" in r.get("Kernel_Name", ""):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = sorted(v)
        print(f.split("/")[-1], k, "launches", len(v), "median", v[len(v) // 2], "min", v[0], "max", v[-1])
PY
