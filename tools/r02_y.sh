#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv_b16.py -x -q -s 2>&1 | grep -E "mask head|passed|failed|Error|error|assert" | head -10
timeout 300 python bench.py --no-cpu-baseline --workload detect --masks 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('split_bf16',{}); print('detect+masks fp32 %.1f M (%.3f ms) | split %.1f M (%.3f ms) single %.3f' % (d['value']/1e6, d['ms_per_step'], s.get('value',0)/1e6, s.get('ms_per_step',0), s.get('single_chunk_latency_ms') or 0))"
timeout 300 python bench.py --no-cpu-baseline --workload images 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('split_bf16',{}); print('images fp32 %.1f M (%.3f ms) | split %.1f M (%.3f ms) single %.3f' % (d['value']/1e6, d['ms_per_step'], s.get('value',0)/1e6, s.get('ms_per_step',0), s.get('single_chunk_latency_ms') or 0))"
