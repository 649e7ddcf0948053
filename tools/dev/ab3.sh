#!/bin/bash
# A/B of two builds of the library on the config 3 bench (alternating, 3 rounds)
for r in 1 2 3; do for l in "$@"; do
  SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/$l timeout 200 python bench.py --config 3 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print('$l', round(d['ms_per_step'],3))"
done; done
