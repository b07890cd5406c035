#!/bin/bash
OUT=gpurun_out/r02s; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_scene.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5
timeout 300 python tools/merge_time.py 32 2>&1 | grep -v Warning | tee $OUT/merge_time.txt
timeout 300 python tools/scene_profile.py 4 32 2>&1 | grep n_chunks | tee $OUT/scene_profile.txt
