#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_conv_t16.py tests/test_gpu_network.py -m gpu -x -q 2>&1 | tail -2
for tg in 0 -1; do
SIS3D_T16_XCD_TG=$tg timeout 300 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tg=$tg', d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['stages']['backbone']['ms'], d['stages']['rpn']['ms'], d['config']['single_chunk_latency_ms'])"
done
