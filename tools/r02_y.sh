#!/bin/bash
for i in 1 2 3 4 5; do
SIS3D_FORCE_DIST=1 timeout 300 python bench.py --workload scene --scene-chunks 4 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['split_bf16']; print('scene4: fp32 %.3f ms | split %.3f ms' % (d['ms_per_step'], s['ms_per_step']))"
done
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['split_bf16']; print('default: fp32 %.1f M | split %.1f M' % (d['value']/1e6, s['value']/1e6))"
