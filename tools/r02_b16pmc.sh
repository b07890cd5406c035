#!/bin/bash
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02_b16pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pb
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pb -- python $ROOT/tools/b16_pmc.py > /tmp/pb.log 2>&1
  f=$(find /tmp/pb -name "*counter_collection.csv" | head -1); t=$(find /tmp/pb -name "*kernel_trace.csv" | head -1)
  python - "$f" "$t" <<'PY' | tee -a $OUT/pmc.txt
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k3b16" in r.get("Kernel_Name", ""):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[2])) if "k3b16" in r["Kernel_Name"])
print("median us %.1f" % dur[len(dur) // 2], {k: sorted(v)[len(v) // 2] for k, v in agg.items()})
PY
done
