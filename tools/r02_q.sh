#!/bin/bash
# HEAD validation on the GPU box: GPU suite, default bench line (stages), scene phase split, per-launch table of one chunk
OUT=gpurun_out/r02q; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 300 python bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read())
print("default", d["value"], d["ms_per_step"], d["config"]["single_chunk_latency_ms"], d["roofline"]["frac"], {k:(v.get("ms") if isinstance(v,dict) else v) for k,v in d.get("stages",{}).items()})
PY
timeout 300 python tools/scene_profile.py 4 32 2>&1 | grep n_chunks | tee $OUT/scene_profile.txt
bash tools/prof_inflight1.sh r02q backbone_rpn
