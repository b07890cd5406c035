#!/usr/bin/env python
"""PMC counter CSVs (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE) -> per-kernel HBM traffic table.

  --reduce <counter_collection.csv>   -> JSON {(kernel, workgroups): {counter: median value, us: median duration}}
  --table  <dir with *_FETCH_SIZE.json / *_WRITE_SIZE.json>  -> the judged table (profiles/r02_hbm_kernels.json)

Units / corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads -> x2.  WRITE_SIZE uncorrected.  Algorithmic bytes: inputs + outputs (+ weights) once, for the
bench's 96x48x96 geometry-only chunk."""
import collections
import csv
import glob
import json
import os
import re
import sys

MB = 1e6
V48, V24 = 48 * 24 * 48, 24 * 12 * 24
# (substring of the kernel name, workgroups or None) -> (label, algorithmic bytes)
ALGO = [
    ("stem_planar_kernel<32, 32>", None, "geometry1[0] planar k2s2 2->32 + conv1 32->32 (pointwise.hip)", 4 * (2 * 96 * 48 * 96 + 2 * V48 * 32)),
    ("pw16_kernel<32, 32, 32, 1, 1>", None, "conv3 32->32 + residual + ReLU + next conv1 @48x24x48", 4 * V48 * 32 * 4),
    ("pw16_kernel<32, 32, 0, 1, 1>", None, "conv3 32->32 + residual + ReLU @48x24x48", 4 * V48 * 32 * 3),
    ("pw16_kernel<32, 128, 32, 4, 8>", None, "k2s2 stem 32->128 + conv1 128->32", 4 * (V48 * 32 + V24 * 128 + V24 * 32 + 8 * 32 * 128)),
    ("pw16_kernel<32, 128, 32, 4, 1>", None, "conv3 32->128 + residual + ReLU + next conv1 @24x12x24", 4 * V24 * (32 + 128 + 128 + 32)),
    ("pw16_kernel<32, 128, 0, 4, 1>", None, "conv3 32->128 + residual + ReLU @24x12x24", 4 * V24 * (32 + 128 + 128)),
    ("pw16_kernel<128, 64, 0, 4, 1>", None, "conv1 128->64 @24x12x24", 4 * V24 * (128 + 64)),
    ("pw16_kernel<64, 128, 64, 4, 1>", None, "conv3 64->128 + residual + ReLU + next conv1 @24x12x24", 4 * V24 * (64 + 128 + 128 + 64)),
    ("pw16_kernel<64, 128, 0, 4, 1>", None, "conv3 64->128 + residual + ReLU @24x12x24", 4 * V24 * (64 + 128 + 128)),
    ("maxpool3_lds_kernel", 1024, "MaxPool3d(3,1,1) 64 ch @48x24x48 (colour path), halo brick through LDS (r4)", 4 * V48 * 64 * 2),
    ("maxpool3_lds_kernel", None, "MaxPool3d(3,1,1) 128 ch @24x12x24, halo brick through LDS (r4)", 4 * V24 * 128 * 2),
    ("conv3d_k3wino_kernel<2, 32, 32", 216, "Bottleneck(32,32) body @48x24x48 on the Winograd kernel + next conv1 (r4): y1 + residual in, out + y1n out",
     4 * (V48 * 32 * 4 + 64 * 32 * 32 + 2 * 32 * 32)),
    ("conv3d_k3wino_kernel<2, 32, 0", 216, "Bottleneck(32,32) body @48x24x48 on the Winograd kernel (r4): y1 + residual in, out",
     4 * (V48 * 32 * 3 + 64 * 32 * 32 + 32 * 32)),
    ("conv3d_k3wino_kernel<2, 128, 0", None, "Bottleneck(128,32) body @24x12x24 on the Winograd kernel (shared-chip form)",
     4 * (V24 * (32 + 128 + 128) + 64 * 32 * 32 + 128 * 32)),
    ("enet_block_kernel<128, 32, 32, 4", None, "ENet bottleneck, stage 2/3 (5 x 32 x 41 px): x 128 + y1 32 in, out 128 + y1n 32, weights once",
     4 * (6560 * (128 + 32 + 128 + 32) + 9 * 32 * 32 + 128 * 32 + 32 * 128)),
    ("enet_block_kernel<64, 16, 16, 4", None, "ENet bottleneck, stage 1 (5 x 64 x 82 px)", 4 * (26240 * (64 + 16 + 64 + 16) + 9 * 16 * 16 + 64 * 16 + 16 * 64)),
    ("proj_tile_kernel", None, "sparse colour stem on the back-projected volume: active output voxels only", None),
    ("upload_kernel", None, "chunk upload by a kernel reading pinned host memory (3.54 MB over PCIe in, 3.54 MB out)", 2 * 4 * 2 * 96 * 48 * 96),
    ("maxpool3_kernel<1>", 865, "MaxPool3d(3,1,1) 128 ch @24x12x24", 4 * V24 * 128 * 2),
    ("maxpool3_kernel<2>", None, "MaxPool3d(3,1,1) 64 ch @48x24x48", 4 * V48 * 64 * 2),
    ("rpn_heads_kernel", None, "both RPN heads of both levels (score, prob, bbox)", 4 * V24 * (2 * 256 + 30 + 110)),
    ("roi_pool_slab_kernel", None, "two-level RoI pooling, 200 rows (live ~100)", None),
    ("decode_kernel", 86, "proposal decode, level 2 (21982 inside anchors)", 21982 * (24 + 24 + 4 + 4 + 32)),
    ("decode_kernel", 45, "proposal decode, level 1 (11412 inside anchors)", 11412 * (24 + 24 + 4 + 4 + 32)),
    ("proj_gather", None, "view-max gather: 226.5 MB volume written once", 4 * 96 * 48 * 96 * 128 + 5 * 1312 * 128 * 4),
    ("proj_table", None, "voxel->pixel table build (5 views)", 5 * 96 * 48 * 96 * 4 + 5 * 3000 * 16),
    ("proj_transpose", None, "feature maps to pixel-major rows", 2 * 5 * 1312 * 128 * 4),
    ("tsdf_encode_kernel", None, "TSDF encode 96x48x96 (4 B in, 8 B out per voxel)", 96 * 48 * 96 * 12),
    ("frustum_kernel", None, "compute_projection, 5 views (packed int64 lists)", None),
    ("fc16_splitk_kernel", None, "classifier fc1 8192->256, split-K (8 MB weights)", 4 * (8192 * 256 + 104 * 8192 + 8 * 112 * 256)),
    ("conv3d_k3t16_kernel<6, 6, 12", 256, "k3 conv on 6x6x12 bricks (rpn_net 128->256: MFMA-bound, for reference)", 4 * (V24 * 128 + V24 * 256 + 27 * 128 * 256)),
]


def short(n):
    return re.sub(r"\(anonymous namespace\)::|void ", "", n)


def reduce_csv(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        key = "%s|%d" % (short(r["Kernel_Name"])[:90], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, d in agg.items():
        out[k] = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
        out[k]["launches"] = len(d["us"])
    json.dump(out, sys.stdout, indent=0)


def table(d):
    merged = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(d, "*_FETCH_SIZE.json")) + glob.glob(os.path.join(d, "*_WRITE_SIZE.json")):
        for k, v in json.load(open(f)).items():
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                if c in v:
                    merged[k][c] = v[c]
            merged[k].setdefault("us", []).append(v["us"])
    rows = []
    for key, v in merged.items():
        name, wgs = key.rsplit("|", 1)
        hit = [a for a in ALGO if a[0] in name and (a[1] is None or a[1] == int(wgs))]
        if not hit:
            continue
        _, _, label, algo = hit[0]
        us = sorted(v["us"])[0]
        fetch = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024.0          # KB, x2 gfx950 correction
        write = v.get("WRITE_SIZE", 0.0) * 1024.0
        rows.append({"kernel": name, "workgroups": int(wgs), "what": label, "us": us,
                     "algorithmic_mb": None if algo is None else algo / MB, "fetch_mb": fetch / MB, "write_mb": write / MB,
                     "pmc_mb": (fetch + write) / MB, "algo_gbs": None if algo is None else algo / us / 1e3,
                     "traffic_ratio": None if not algo else (fetch + write) / algo,
                     "pmc_gbs": (fetch + write) / us / 1e3, "hbm_frac": None if algo is None else algo / us / 1e3 / 8000.0,
                     "hbm_frac_pmc": (fetch + write) / us / 1e3 / 8000.0})
    rows.sort(key=lambda r: -r["us"])
    json.dump({"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --workload detect "
                      "--inflight 1 --no-graph` and tools/hbm_drivers.py; medians per (kernel, grid); FETCH_SIZE x2 (gfx950), KB -> B; "
                      "us = shortest median of the two passes; peak 8 TB/s; working sets of 3.5-28 MB sit in the 256 MB Infinity Cache, "
                      "whose hits the counters include (MI355X_MICROARCH.md)", "kernels": rows}, sys.stdout, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "--reduce":
        reduce_csv(sys.argv[2])
    else:
        table(sys.argv[2])
