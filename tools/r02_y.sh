#!/bin/bash
for n in 2 3 4 5 6; do
timeout 300 python bench.py --no-cpu-baseline --no-stages --inflight $n --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('split_bf16',{}); print('inflight $n: fp32 %.1f M | split %.1f M' % (d['value']/1e6, s.get('value',0)/1e6))"
done
