#!/bin/bash
timeout 300 python bench.py --no-cpu-baseline 2> /tmp/b.err | tail -1 > gpurun_out/bench_split.json; tail -3 /tmp/b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_split.json"))
print("fp32  value %.1f M  ms %.3f single %.3f" % (d["value"]/1e6, d["ms_per_step"], d["config"]["single_chunk_latency_ms"]))
s=d.get("split_bf16")
print("split", json.dumps(s)[:900] if s else None)
PY
