#!/bin/bash
# streamed-input variants of bench.py's chunk_pipeline / scene keys: upload by kernel vs hipMemcpyAsync, calibrated window vs default
F="--gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-live-pmc --no-side-configs --no-stages"
for v in "kernel:" "own:" "own:--no-calibrate" "kernel:--no-calibrate"; do
  c=${v%%:*}; x=${v#*:}
  echo "== SIS3D_FEED_COPY=$c $x"
  SIS3D_FEED_COPY=$c python bench.py $F $x 2>/dev/null | python tools/show_line.py | grep -v "^detect\|^images\|calibration\|stream window"
done
