#!/bin/bash
# HBM traffic of the rpn_net launch, default work-list order vs XCD ranges of bricks x tiles (FETCH_SIZE / WRITE_SIZE, separate passes)
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tg in 0 -1; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/px
  SIS3D_T16_XCD_TG=$tg timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/px -- python $ROOT/tools/t16_pmc.py rpn 1 > /tmp/px.log 2>&1
  f=$(find /tmp/px -name "*counter_collection.csv" | head -1)
  python - "$f" $tg $ctr <<'PY' | tee -a $OUT/traffic.txt
import csv, sys
v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k3t16" in r.get("Kernel_Name", ""))
print("tg=%s %s launches %d median %.1f KB" % (sys.argv[2], sys.argv[3], len(v), v[len(v) // 2]))
PY
done
done
