#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv_b16.py -x -q 2>&1 | tail -1
timeout 200 python tools/b16_phases.py 3 2>&1 | grep -v -i warn | grep -E "brick|chunk|prologue|reduction"
timeout 300 python tools/b16_time.py 2>&1 | grep " us " | grep -E "rpn|geometry"
