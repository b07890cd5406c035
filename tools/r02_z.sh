#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
bash tools/r02_profiles.sh r02_final4 2>&1 | grep "^bench_"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
