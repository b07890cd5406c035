#!/bin/bash
for wl in backbone_rpn detect; do
timeout 300 python bench.py --no-cpu-baseline --workload $wl 2> /tmp/b.err | tail -1 > gpurun_out/bench_split_$wl.json; tail -2 /tmp/b.err
python - $wl <<'PY'
import json, sys
d=json.load(open("gpurun_out/bench_split_%s.json" % sys.argv[1]))
s=d.get("split_bf16") or {}
print(sys.argv[1], "fp32 %.1f M (%.3f ms, single %.3f) | split %.1f M (%.3f ms, single %.3f) stages %s" % (d["value"]/1e6, d["ms_per_step"], d["config"]["single_chunk_latency_ms"], s.get("value",0)/1e6, s.get("ms_per_step",0), s.get("single_chunk_latency_ms",0), {k:round(v["ms"],4) for k,v in s.get("stages",{}).items()}))
PY
done
timeout 300 python tools/b16_time.py 2>&1 | grep " us " > gpurun_out/b16_time.txt; cat gpurun_out/b16_time.txt | head -3
