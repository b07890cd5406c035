#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do SIS3D_WINO_WC=1 timeout 60 tools/_bin/wino_bench_0 128 256 24 12 24 1 | sed "s/^/wc=1 /"; done
for i in 1 2; do SIS3D_WINO_WC=2 timeout 60 tools/_bin/wino_bench_0 128 256 24 12 24 1 | sed "s/^/wc=2 /"; done
