#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/dbg
timeout 300 python -m pytest tests/test_gpu_conv_wino.py -x -q -m gpu > gpurun_out/dbg/wino.txt 2>&1; echo "wino rc $?"
timeout 300 python -m pytest tests/test_gpu_bottleneck.py -x -q -m gpu -k "not shared and not 128_channels" > gpurun_out/dbg/bn_old.txt 2>&1; echo "bn_old rc $?"
timeout 300 python -m pytest tests/test_gpu_bottleneck.py -x -v -m gpu -k "shared_chip_sends" > gpurun_out/dbg/bn_shared.txt 2>&1; echo "bn_shared rc $?"
timeout 300 python -m pytest tests/test_gpu_bottleneck.py -x -v -m gpu -k "128_channels" > gpurun_out/dbg/bn_128.txt 2>&1; echo "bn_128 rc $?"
for f in gpurun_out/dbg/*.txt; do echo "== $f"; grep -v "^  File\|^Extension" $f | head -40; done
