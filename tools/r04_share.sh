#!/bin/bash
# a 4-chunk scene (= what one rank of 8 owns of the 32-chunk scene) with 2 / 3 / 4 pipelines: 4 = one graph launch for the round
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for n in 4 3 2; do
SIS3D_FORCE_DIST=1 python bench.py --workload scene --scene-chunks 4 --inflight $n --steps 100 --warmup 20 --no-cpu-baseline --no-side-workloads --no-split-line --no-stages --no-live-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4-chunk scene, pipelines $n: ms/scene %.4f' % d['ms_per_step'], d['config'].get('one_graph_launch_per_scene'))"
done; done
