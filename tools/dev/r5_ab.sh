#!/bin/bash
# A/B of library variants on one box: tools/dev/r5_ab.sh <config> <steps> <reps> lib1 lib2 ...   (names under scikit-downscale_amd/lib)
set -u
cfg=$1; steps=$2; reps=$3; shift 3
L=$PWD/scikit-downscale_amd/lib
for i in $(seq $reps); do
for lib in "$@"; do
  SD_DOWNSCALE_LIB=$L/$lib timeout 300 python bench.py --config $cfg --steps $steps --warmup 3 --parity-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('%-28s c$cfg %7.3f ms/step  kernels %7.3f ms  frac %.4f  %s parity %s' % ('$lib', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], {k: round(v*r['launches_per_step'][k],3) for k,v in r['per_kernel_avg_ms'].items() if v*r['launches_per_step'][k] > 0.05}, d.get('parity_check')))"
done; done
