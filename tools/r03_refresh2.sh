#!/bin/bash
# the remaining bench lines on the final code (after the launch-bounds fix of the 3x3x3 Bottleneck body)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_final2; mkdir -p $O
run() { timeout 500 "$@"; }
run python bench.py --workload detect --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > "$O/bench_detect.json"
run python bench.py --workload detect --masks --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > "$O/bench_detect_masks.json"
run python bench.py --workload images --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > "$O/bench_images.json"
run python bench.py --workload images --rgb --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > "$O/bench_images_rgb.json"
SIS3D_FORCE_DIST=1 run python bench.py --workload scene --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > "$O/bench_scene.json"
SIS3D_FORCE_DIST=1 run python bench.py --workload scene --scene-chunks 4 --steps 100 --warmup 20 --no-cpu-baseline --no-side-workloads 2>/dev/null | grep "^{" | tail -1 > "$O/bench_scene4.json"
for f in $O/bench_detect.json $O/bench_detect_masks.json $O/bench_images.json $O/bench_images_rgb.json $O/bench_scene.json $O/bench_scene4.json; do python -c "
import json; d=json.loads(open('$f').read()); c=d['config']; print('$f', round(d['value']/1e9,4), round(d['ms_per_step'],4), c.get('single_chunk_latency_ms'), {k:v for k,v in c.items() if 'mask_head' in k or 'enet_ms' in k}, {k:d[k] for k in d if 'mask_head' in k})"; done
