#!/bin/bash
# weight_analogs at BASELINE size against the window length (slope = cost of the window reads, intercept = fill + search)
run() { timeout 600 python bench.py --config 4 --parity-only --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-60s %9.0f cells/s %8.2f ms parity %s' % (d['config']['workload'][:60], d['value'], d['ms_per_step'], d['parity_check']))
print('      ', {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if v*r['launches_per_step'][k] > 0.3})"; }
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/${1:-libsd_downscale.so}
for k in 2 8 16 30 60; do run --analog-kind weight_analogs --analog-k $k; done
for k in 4 16 30 60; do run --analog-estimator regression --analog-k $k; done
