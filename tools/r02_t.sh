#!/bin/bash
OUT=$PWD/gpurun_out/r02t; mkdir -p $OUT
ROOT=$PWD
timeout 300 python -m pytest tests/test_gpu_scene.py -x -q -k fused 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- python $ROOT/tools/merge_time.py 32 > /tmp/pm.log 2>&1
grep -E "fused|merge_scene" /tmp/pm.log
f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/merge_kernels.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("scene_", "nms_cand", "nms_resolve", "fillBuffer", "memset")):
        print("%-40s calls %4s avg %8.1f us" % (n.split("(")[1][:40] if n.startswith("void (") or n.startswith("(") else n[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
