#!/bin/bash
# analog surface at BASELINE size: kinds, AnalogRegression, F = 3 (one line each)
L=${1:-libsd_downscale.so}
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/$L
run() { timeout 600 python bench.py --config 4 --parity-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-70s %9.0f cells/s %8.2f ms frac %.4f parity %s' % (d['config']['workload'][:70], d['value'], d['ms_per_step'], r['frac'], d['parity_check']))
print('      ', {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if v*r['launches_per_step'][k] > 0.3})"; }
run --steps 6 --warmup 2
run --steps 6 --warmup 1 --analog-kind best_analog --analog-k 200
run --steps 3 --warmup 1 --analog-kind weight_analogs
run --steps 3 --warmup 1 --analog-estimator regression
run --steps 2 --warmup 1 --analog-features 3 --cells 16384
