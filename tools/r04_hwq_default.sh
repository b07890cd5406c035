#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for q in 4 8; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-line --no-live-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scene']; print('hwq=$q value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'chunks/step', d['config']['chunks_per_step_per_gpu'], 'alone %.4f' % d['config']['single_chunk_latency_ms'], '| scene %.3f ms, share %.4f ms, ceiling %.2f, streams %s' % (s['ms_per_scene'], s['share_of_one_rank_at_8']['ms'], s['share_of_one_rank_at_8']['ceiling_speedup_at_8'], s['streams_per_gpu']))"
done; done
