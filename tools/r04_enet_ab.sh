#!/bin/bash
# A/B of the ENet bottleneck kernel's two forms: SIS3D_ENET_WAVES=1 (one wave per workgroup, weights from L2) vs 4 (weights in LDS)
for w in 1 4 1 4; do SIS3D_ENET_WAVES=$w python bench.py --workload images --rgb --no-cpu-baseline --no-side-workloads --no-split-line 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves', $w, 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'alone %.3f' % d['config']['single_chunk_latency_ms'], 'enet_ms_5_views %.4f' % d['config']['enet_ms_5_views'])"; done
