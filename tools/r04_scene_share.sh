for st in 1 0; do SIS3D_ROUND_STAGGER=$st python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-line --no-stages 2>gpurun_out/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scene']; print('stagger', $st, s['ms_per_scene'], s['share_of_one_rank_at_8']['ms'], s['share_of_one_rank_at_8']['ceiling_speedup_at_8'])"; done
