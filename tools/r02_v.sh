#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_mask_variants.py tests/test_gpu_conv_t16.py -m gpu -x -q 2>&1 | tail -2
timeout 500 python tools/mask_time.py 16 2>&1 | grep -E "^brick -1"
timeout 300 python bench.py --no-cpu-baseline --steps 100 --workload detect --masks 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('detect+masks', d['value'], d['ms_per_step'])"
