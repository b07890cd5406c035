#!/bin/bash
# fused Bottleneck: parity, then bench A/B (one launch vs k3t16 + pointwise)
mkdir -p gpurun_out/r02p
timeout 600 python -m pytest tests/test_gpu_bottleneck.py -x -q 2>&1 | tail -15
for mode in fused split; do
  if [ $mode = split ]; then export SIS3D_BNECK_SPLIT=1; else unset SIS3D_BNECK_SPLIT; fi
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02p/bench_$mode.json 2> gpurun_out/r02p/bench_$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r02p/bench_$mode.json").read().strip().splitlines()[-1])
print("$mode", d["value"], d["ms_per_step"], {k:(v.get("ms") if isinstance(v,dict) else v) for k,v in d.get("stages",{}).items()}, d["roofline"]["achieved"])
PY
done
