#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for b in old 0; do for a in "128 256 24 12 24 1" "128 256 24 12 24 2" "32 32 48 24 48 1"; do echo -n "$b: "; timeout 60 tools/_bin/wino_bench_$b $a | head -1; done; done; done
