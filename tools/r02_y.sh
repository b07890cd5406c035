#!/bin/bash
SIS3D_FORCE_DIST=1 timeout 300 python bench.py --workload scene --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_scene_split.json
SIS3D_FORCE_DIST=1 timeout 300 python bench.py --workload scene --scene-chunks 4 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_scene4_split.json
python - <<'PY'
import json
for n in ("scene","scene4"):
    d=json.load(open("gpurun_out/bench_%s_split.json"%n)); s=d["split_bf16"]
    print(n, "fp32 %.3f ms %.0f M | split %.3f ms %.0f M records %d kept %d (fp32 kept %d)" % (d["ms_per_step"], d["value"]/1e6, s["ms_per_step"], s["value"]/1e6, s["records_gathered"], s["kept_after_scene_nms"], d["config"]["kept_after_scene_nms"]))
PY
