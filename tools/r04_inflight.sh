#!/bin/bash
# chunks in flight for the headline workload with the round-4 kernel mix (Winograd Bottleneck bodies take their CUs whole)
for n in 3 4 5 6 8; do python bench.py --inflight $n --steps 30 --warmup 5 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', $n, 'value %.4g' % d['value'], 'ms/chunk %.4f' % (d['ms_per_step'] / $n))"; done
for n in 2 3 4; do python bench.py --workload detect --inflight $n --steps 30 --warmup 5 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('detect inflight', $n, 'value %.4g' % d['value'], 'ms/chunk %.4f' % (d['ms_per_step'] / $n))"; done
