#!/bin/bash
# streamed variants through the product API (bench.py's chunk_pipeline + scene keys), by copy mode and hardware-queue count
for q in 8 16; do for c in own per_pipeline; do
  echo "== GPU_MAX_HW_QUEUES=$q SIS3D_FEED_COPY=$c"
  GPU_MAX_HW_QUEUES=$q SIS3D_FEED_COPY=$c python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-live-pmc --no-side-configs 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
cp, sc = d['chunk_pipeline'], d['scene']
s = sc['share_of_one_rank_at_8']
print('resident %.3f ms  grid %.3f (%.3f)  sdf %.3f (%.3f) | scene %.2f ms streamed %.2f (%.3f) | share %.3f ms ceiling %.2f' % (
    cp['ms_per_step'], cp['streamed']['ms_per_step'], cp['streamed']['ratio_to_resident'], cp['streamed_sdf']['ms_per_step'],
    cp['streamed_sdf']['ratio_to_resident'], sc['ms_per_scene'], sc['streamed']['ms_per_scene'], sc['streamed']['ratio_to_resident'],
    s['ms'], s['ceiling_speedup_at_8']))
"
done; done
