#!/bin/bash
# where in a K-step the eight LDS-DMA pieces of the U stage go: first virtual slot _ stride between pieces
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do for b in tools/_bin/wino_fill_*; do echo -n "$(basename $b): "; timeout 60 $b 128 256 24 12 24 1 | head -1; done; done
for b in tools/_bin/wino_fill_*; do echo -n "$(basename $b): "; timeout 60 $b 128 256 24 12 24 1 | tail -1 | cut -c1-50; echo -n "$(basename $b) NC=1: "; timeout 60 $b 128 128 24 12 24 1 | head -1;  done
