cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final2; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_bottleneck.py -x -q --tb=line 2>&1 | tail -1
python bench.py 2>/dev/null | grep "^{" | tail -1 > $O/bench_backbone_rpn.json
python bench.py --inflight 1 --no-cpu-baseline --no-side-workloads 2>/dev/null | grep "^{" | tail -1 > $O/bench_backbone_rpn_inflight1.json
python bench.py --inflight 3 --no-cpu-baseline --no-side-workloads 2>/dev/null | grep "^{" | tail -1 > $O/bench_backbone_rpn_inflight3.json
cd /tmp && export TMPDIR=/tmp
for wl in backbone_rpn; do
  rm -rf /tmp/prof_$wl
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-side-workloads > /tmp/prof_$wl.log 2>&1
  grep "^{" /tmp/prof_$wl.log | tail -1 > "$GRAFT_REPO_ROOT/$O/bench_${wl}_under_rocprof.json"
  f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$GRAFT_REPO_ROOT/$O/bench_${wl}_kernel_stats.csv"
  t=$(find /tmp/prof_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$GRAFT_REPO_ROOT/tools/trace_by_grid.py" "$t" > "$GRAFT_REPO_ROOT/$O/bench_${wl}_by_grid.md"
  python "$GRAFT_REPO_ROOT/tools/dominant_from_trace.py" "$t" "$GRAFT_REPO_ROOT/$O/bench_${wl}_under_rocprof.json" > "$GRAFT_REPO_ROOT/$O/dominant_kernel_from_trace.json"
  python "$GRAFT_REPO_ROOT/tools/dominant_from_trace.py" --direct "$t" > "$GRAFT_REPO_ROOT/$O/direct_kernel_from_trace.json"
  rm -rf /tmp/prof1_$wl
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$wl -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-stages --no-side-workloads --no-split-line > /tmp/prof1_$wl.log 2>&1
  t=$(find /tmp/prof1_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$GRAFT_REPO_ROOT/tools/trace_by_grid.py" "$t" > "$GRAFT_REPO_ROOT/$O/bench_${wl}_inflight1_by_grid.md"
done
cd $GRAFT_REPO_ROOT
for f in $O/bench_backbone_rpn.json $O/bench_backbone_rpn_inflight1.json $O/bench_backbone_rpn_inflight3.json; do python -c "
import json,sys; d=json.loads(open('$f').read()); print('$f', round(d['value']/1e9,4), round(d['ms_per_step'],4), d['config'].get('single_chunk_latency_ms'), {k:round(v['ms'],4) for k,v in d.get('stages',{}).items() if isinstance(v,dict)}, round(d['roofline']['launch_us'],2))"; done
grep "bottleneck16" $O/bench_backbone_rpn_inflight1_by_grid.md | cut -c1-120
