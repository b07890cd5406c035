python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "roi" 2>&1 | tail -1
for i in 1 2 3; do python bench.py --workload detect --steps 20 --warmup 5 2>gpurun_out/err_$i.txt | python -c "
import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print(l['value']/1e9, l['ms_per_step'], l.get('stream_placement') or l['config'].get('stream_placement'))"; done
