#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv_b16.py -x -q 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('split_bf16',{}); print('fp32 %.1f M | split %.1f M (%.3f ms) single %.3f stages %s' % (d['value']/1e6, s.get('value',0)/1e6, s.get('ms_per_step',0), s.get('single_chunk_latency_ms') or 0, {k:round(v['ms'],4) for k,v in s.get('stages',{}).items()}))"
