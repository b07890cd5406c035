#!/bin/bash
# chunks in flight per GPU x workload, with the stream window chosen by measurement (PipelinedEngines.calibrate)
F="--gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-live-pmc --no-side-workloads --no-stages --no-streamed"
for w in backbone_rpn detect images; do for n in 4 5 6; do
  printf "%-13s inflight %d: " $w $n
  python bench.py $F --workload $w --inflight $n 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%.3f G voxels/s  %.3f ms/step  window %s' % (d['value'] / 1e9, d['ms_per_step'], d['config'].get('stream_window')))
"
done; done
