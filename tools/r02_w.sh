#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -x -q -k "fork or full_forward" 2>&1 | tail -3
for wl in backbone_rpn detect; do
timeout 300 python bench.py --no-cpu-baseline --steps 100 --workload $wl 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$wl', d['value'], d['roofline']['launch_us'], c['single_chunk_latency_ms'], c.get('single_chunk_latency_forked_ms'))"
done
