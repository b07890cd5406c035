#!/bin/bash
for rep in 1 2; do for n in 3 4 6; do python bench.py --inflight $n --steps 200 --warmup 5 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', $n, 'value %.4g' % d['value'], 'ms/chunk %.4f' % (d['ms_per_step'] / $n))"; done; done
