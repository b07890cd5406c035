#!/usr/bin/env python
"""rocprofv3 kernel-trace CSV -> durations of the dominant conv template split by launch grid.
The template conv3d_mfma_kernel<3,1,4,4,4,2,2,3,1,32,false,...> serves geometry2.0 (216 workgroups), ONE 128->256 RPN conv
(432 workgroups: the launches bench.py times for `roofline.achieved`) and the batched pair of RPN convs (864)."""
import collections
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv3d_mfma_kernel<3, 1, 4, 4, 4, 2, 2, 3, 1, 32, false" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"kernel": "conv3d_mfma_kernel<3,1,4,4,4,2,2,3,1,32,false,4,false>", "by_workgroups": {}}
for g, v in sorted(by.items()):
    out["by_workgroups"][str(g)] = {"launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
d = out["by_workgroups"].get("432")
if d:
    out["dominant_single_rpn_conv"] = {"avg_us": d["avg_us"], "tflops": 2.0 * 6912 * 256 * 128 * 27 / d["avg_us"] / 1e6,
                                       "note": "432-workgroup launches = bench.py's timing launches of the dominant kernel"}
print(json.dumps(out, indent=1))
