#!/bin/bash
# share_of_one_rank_at_8 with the merge on its own stream vs on the caller's stream
for m in 1 0; do
  echo "== SIS3D_MERGE_STREAM=$m"
  SIS3D_MERGE_STREAM=$m python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --no-side-configs --no-streamed --no-stages 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s = d['scene']['share_of_one_rank_at_8']
print('value %.3f G  scene %.3f ms  share %.3f ms  ceiling %.2f' % (d['value'] / 1e9, d['scene']['ms_per_scene'], s['ms'], s['ceiling_speedup_at_8']))
"
done
