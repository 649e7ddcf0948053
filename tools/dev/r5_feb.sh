#!/bin/bash
# config 2: whole-lane months in their own launch (default) against one launch of the general kernel for all months
L=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
run() { SD_DOWNSCALE_LIB=$L timeout 300 python bench.py --config 2 --steps 20 --warmup 3 --parity-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('%-28s %7.3f ms/step  kernels %7.3f ms  frac %.4f  %s parity %s' % ('$1', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], {k: round(v*r['launches_per_step'][k],3) for k,v in r['per_kernel_avg_ms'].items() if v*r['launches_per_step'][k] > 0.05}, d.get('parity_check')))"; }
run default
SD_FX_NOFULL=1 run nofull
run default
SD_FX_NOFULL=1 run nofull
