#!/usr/bin/env python
"""rocprofv3 kernel_trace.csv -> markdown table of (kernel, grid, workgroup) groups: calls, mean / min / max us.
The stats CSV averages every launch of one template instantiation; layers of different size share instantiations."""
import collections
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    agg = collections.defaultdict(list)
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
        gy = int(r.get("Grid_Size_Y", 1) or 1) // max(1, int(r.get("Workgroup_Size_Y", 1) or 1))
        key = (n[:100] + (" [x%d problems]" % gy if gy > 1 else ""), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]),
               r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values()) or 1.0
    print("| kernel | workgroups | threads | vgpr | lds | calls | mean us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("| %s | %d | %d | %s | %s | %d | %.1f | %.1f | %.1f | %.1f |" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v), min(v), max(v),
                                                                          100.0 * sum(v) / tot))


if __name__ == "__main__":
    main()
