#!/usr/bin/env python
"""Rebuild the head of profiles/README.md (headline table + per-kernel tables) from profiles/r01_*; the hand-written part from
'## Optimisation log' on is kept."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def table(f, top=14):
    rows = list(csv.DictReader(open(f)))
    mp = sum(int(r["Calls"]) for r in rows if "maxpool3_kernel" in r["Name"])
    chunks = mp / (3 if "images" in f else 1)
    out = ["| kernel | launches/chunk | µs/chunk | mean µs | % |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:95]
        out.append("| `%s` | %.1f | %.1f | %.1f | %.1f |" % (name, int(r["Calls"]) / chunks, float(r["TotalDurationNs"]) / 1e3 / chunks,
                                                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return "\n".join(out)


J = {k: json.load(open(os.path.join(P, "r01_bench_%s.json" % k)))
     for k in ("backbone_rpn", "backbone_rpn_inflight1", "detect", "images", "images_from_depth", "scene")}
T = json.load(open(os.path.join(P, "r01_dominant_kernel_from_trace.json")))


def row(label, k):
    d = J[k]
    c = d["config"]
    lat = ("%.3f ms" % c["single_chunk_latency_ms"]) if c.get("single_chunk_latency_ms") else "—"
    return "| %s | %.3f | **%.0f M** | %s | %.1f %% | %.1f %% |" % (label, d["ms_per_step"], d["value"] / 1e6, lat,
                                                                100 * d["step_roofline"]["fp32_frac"], 100 * d["step_roofline"]["hbm_frac"])


s = open(os.path.join(P, "README.md")).read()
tail = s[s.index("## Optimisation log"):]
d0, i1 = J["backbone_rpn"], J["backbone_rpn_inflight1"]
cb = d0["cpu_baseline"]
tr = T["dominant_single_rpn_conv"]
head = f"""# profiles/ — round 1 measurements (one MI355X, gfx950, ROCm 7.2, `gpurun`)

All numbers: synthetic 96x48x96 chunks, seeded synthetic weights, fp32, inputs resident in HBM, HIP-graph replay.
Files: `r01_bench_*.json` = the bench.py JSON lines; `r01_*_kernel_stats.csv` = `rocprofv3 --kernel-trace --stats
--output-format csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline [--workload ...]` (kernel_stats only;
the traces stay in gpurun_out/); `r01_dominant_kernel_from_trace.json` = the dominant kernel's launches of that same trace split by
grid size (tools/dominant_from_trace.py); `r01_pmc_rpn_net.json` / `r01_pmc_rpn_net_mfma.json` = HBM traffic and MFMA-busy counters of the dominant kernel (separate --pmc passes).
Everything here is produced by `tools/round_profiles.sh` (one `gpurun` call) and this file's head by `tools/make_profiles_readme.py`.

## Headline (bench.py defaults: backbone + RPN, 3 chunks in flight per GPU)

| workload | ms / step (3 chunks) | voxels/s | single-chunk latency | fp32-roof frac of step | HBM-roof frac of step |
|---|---|---|---|---|---|
{row('backbone + RPN (config 1) — the bench default', 'backbone_rpn')}
| same, 1 chunk in flight | {i1['ms_per_step']:.3f} (1 chunk) | {i1['value'] / 1e6:.0f} M | — | {100 * i1['step_roofline']['fp32_frac']:.1f} % | {100 * i1['step_roofline']['hbm_frac']:.1f} % |
{row('detect = + decode/top-k/NMS/RoI pool/classifier + record packing (config 2 w/o masks)', 'detect')}
{row('images = 5-view back-projection + colour/geometry backbone + RPN (config 3), lists loaded', 'images')}
{row('images, views given as depth maps + poses (visibility lists computed on device inside the step)', 'images_from_depth')}
| scene = 32 chunks, chunk-DP with the RCCL all-gather + whole-scene NMS (config 5), 1 GPU | {J['scene']['ms_per_step']:.2f} (32 chunks) | {J['scene']['value'] / 1e6:.0f} M | — | — | — |

CPU baseline (same box, host cores): oracle backbone+RPN on torch-CPU/oneDNN, {cb['cores']} threads (best of a sweep; the box has 256 cores):
**{cb['value'] / 1e6:.2f} M voxels/s** -> GPU/CPU = {d0['value'] / cb['value']:.0f}x (reported, not a target; 8.8-10.9 M across runs of the same sweep).

Dominant kernel (rpn_net k3 128->256, 12.23 GFLOP/launch): {d0['roofline']['launch_us']:.1f} us -> {d0['roofline']['achieved']:.1f} TFLOP/s = {100 * d0['roofline']['frac']:.0f} % of the 157.3 TF fp32 MFMA roof
(bench.py: mean of 100 warm launches on the launch stream, HIP events).  rocprofv3 trace of the same command, the same 432-workgroup
launches: {tr['avg_us']:.1f} us average ({tr['tflops']:.1f} TFLOP/s) -- agrees within {abs(tr['avg_us'] / d0['roofline']['launch_us'] - 1) * 100:.0f} %.  (The kernel_stats row of this
template averages three different layers: {', '.join('%s workgroups %.0f us' % (k, v['avg_us']) for k, v in T['by_workgroups'].items())}.)
MFMA-pipe utilisation from PMC counters (`r01_pmc_rpn_net_mfma.json`, tools/pmc_mfma.sh): SQ_VALU_MFMA_BUSY_CYCLES = 191,102,976 =
2,985,984 MFMAs x 64 cycles (exactly the algorithmic count) over 300,062 active cycles x 1024 SIMDs = 62 % busy under the profiler.
HBM traffic of that launch (PMC, corrected): 29.7 MB vs 14.2 MB algorithmic -> 0.25 TB/s: HBM is idle, the kernel is MFMA-pipe-bound
(see DESIGN.md section 3 for why 100 % is out of reach at this problem size: 1728 output tiles over 1024 SIMDs).

Progress inside round 1 (same box class): backbone+RPN 903 -> 932 (ragged mask head, fused softmax) -> 949 (max-pool) -> **1.0 G voxels/s**
(third stream; run-to-run 988-1006 M); detect 789 -> 874 -> 884 (one-kernel record packing) -> 907-920 (serialised load chains removed from
the classifier tail, fused stages, NMS select, RoI pooling); images 575 -> 664-684 (colour stem reads the views through the voxel->pixel
table: no 226 MB volume); scene 21.9 -> 18.7-19.0 ms; one chunk alone 0.61 -> 0.58 ms.

### Per-kernel time, backbone+RPN, 3 chunks in flight (kernel durations OVERLAP across the three streams, so the per-chunk column sums to more than the step; the rpn conv row includes the 100 timing launches)

{table(os.path.join(P, 'r01_bench_backbone_rpn_kernel_stats.csv'))}

### Per-kernel time, detect workload, 3 chunks in flight

{table(os.path.join(P, 'r01_bench_detect_kernel_stats.csv'), 22)}

### Per-kernel time, images workload, 3 chunks in flight (`<2, 2, ..., true, 4, true>` is the colour stem reading the projected views)

{table(os.path.join(P, 'r01_bench_images_kernel_stats.csv'), 16)}

Un-overlapped kernel times of the images path (1 chunk in flight, taken before the LDS-cached table lookup went in) are in
`r01_bench_images_inflight1_prefix_kernel_stats.csv`.

Other kernels measured this round (tests/perf_frustum_vs_oracle.py, tools/pool_time.py): `sis3d_compute_projection`, 5 views: 26 us per 96x48x96 chunk
(1.3 TB/s of list writes; torch-CPU oracle 0.80 s), 214 us for a 256x96x320 scene grid (2.9 TB/s; oracle 1.9 s).  max-pool 3x3x3: 6.8 us
(24x12x24x128; 9.8 us before its tap loads were batched), 21.5 us (48x24x48x64,
L1/L2-bound on its 18 taps per output: a separable LDS version is the next step there).

"""
open(os.path.join(P, "README.md"), "w").write(head + tail)
print("profiles/README.md rebuilt")
