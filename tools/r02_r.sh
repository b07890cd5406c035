#!/bin/bash
OUT=gpurun_out/r02r; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k nms 2>&1 | tail -5
timeout 300 python tools/merge_time.py 32 2>&1 | grep -v Warning | tee $OUT/merge_time.txt
