#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2
bash tools/r02_profiles.sh r02_final5 2>&1 | grep "^bench_"
timeout 300 python tools/b16_time.py 2>&1 | grep " us " > gpurun_out/r02_final5/b16_time.txt
timeout 200 python tools/b16_phases.py 3 2>&1 | grep -v -i warn | grep -v amdgpu > gpurun_out/r02_final5/b16_phases.txt
python - <<'PY'
import json
for n in ("backbone_rpn","detect","detect_masks","images","scene","scene4"):
    d=json.load(open("gpurun_out/r02_final5/bench_%s.json"%n)); s=d.get("split_bf16") or {}
    print("SPLIT %-14s fp32 %.1f M (%.3f ms) | split %.1f M (%.3f ms) single %s stages %s" % (n, d["value"]/1e6, d["ms_per_step"], s.get("value",0)/1e6, s.get("ms_per_step",0), s.get("single_chunk_latency_ms"), {k:round(v["ms"],4) for k,v in s.get("stages",{}).items()}))
PY
