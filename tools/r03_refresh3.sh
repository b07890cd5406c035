#!/bin/bash
# after the sparse colour stem: the image-path bench lines, the one-chunk kernel table of the image path, the full GPU suite
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_final2; mkdir -p $O gpurun_out
export SIS3D_PARITY_LOG=$PWD/gpurun_out/r03_parity_log.txt; rm -f $SIS3D_PARITY_LOG
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -2
unset SIS3D_PARITY_LOG
timeout 500 python bench.py --workload images --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > "$O/bench_images.json"
timeout 500 python bench.py --workload images --rgb --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > "$O/bench_images_rgb.json"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- python $GRAFT_REPO_ROOT/bench.py --workload images --inflight 1 --steps 60 --warmup 10 --no-cpu-baseline --no-stages --no-side-workloads --no-split-line > /tmp/pi.log 2>&1
t=$(find /tmp/pi -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $t > $GRAFT_REPO_ROOT/$O/bench_images_inflight1_by_grid.md
cd $GRAFT_REPO_ROOT
for f in $O/bench_images.json $O/bench_images_rgb.json $O/bench_images_dense_stem.json; do python -c "
import json; d=json.loads(open('$f').read()); c=d['config']; print('$f', round(d['value']/1e9,4), round(d['ms_per_step'],4), c.get('single_chunk_latency_ms'), {k:v for k,v in c.items() if 'enet_ms' in k})"; done
grep "proj_\|conv3d_mfma" $O/bench_images_inflight1_by_grid.md | cut -c1-150
