#!/bin/bash
# XCD work order of the rpn_net layer: one cout group x all blocks per XCD (0) against two groups x half of the blocks (1): time and HBM traffic
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
for rep in 1 2 3; do for v in 0 1; do for a in "128 256 24 12 24 1" "128 256 24 12 24 2"; do echo -n "xcd_pairs=$v: "; timeout 60 tools/_bin/wino_xcd_$v $a | tr '\n' ' ' | cut -c1-150; echo; done; done; done
bash tools/r03_wpmc.sh r04_xcd/wino_pmc rpn 2>&1 | tail -12
