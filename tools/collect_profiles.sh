#!/bin/bash
# copy what is judged from gpurun_out/<tag>/ (tools/round_profiles.sh) into profiles/ under the round's prefix
#   usage: bash tools/collect_profiles.sh r06
set -eu
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/$TAG
P=$ROOT/profiles
python "$ROOT/tools/hbm_table.py" --table "$S/hbm" > "$S/hbm/hbm_kernels.json"
cp "$S/bench_driver_style.json" "$P/${TAG}_bench_driver_style.json"
cp "$S/bench_n2_shared_gpu_functional.json" "$P/${TAG}_bench_n2_shared_gpu_functional.json"
for n in backbone_rpn backbone_rpn_inflight1 detect detect_inflight1 detect_masks_inflight1 images_rgb_inflight1; do
  cp "$S/bench_${n}_by_grid.md" "$P/${TAG}_bench_${n}_by_grid.md"
done
cp "$S/bench_backbone_rpn_kernel_stats.csv" "$P/${TAG}_bench_backbone_rpn_kernel_stats.csv"
cp "$S/bench_backbone_rpn_under_rocprof.json" "$P/${TAG}_bench_backbone_rpn_under_rocprof.json"
cp "$S/dominant_kernel_from_trace.json" "$P/${TAG}_dominant_kernel_from_trace.json"
cp "$S/parity_log.txt" "$P/${TAG}_parity_log.txt"
cp "$S/wino_pmc/pmc_summary.json" "$P/${TAG}_pmc_rpn_net_winograd.json"
cp "$S/hbm/hbm_kernels.json" "$P/${TAG}_hbm_kernels.json"
ls "$P" | grep "^${TAG}_" | wc -l
