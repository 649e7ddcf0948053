#!/bin/bash
# round 6: the probability plane left to the staging transpose in the single-pass kernel too (weight_analogs, AnalogRegression without a threshold)
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
run() {  # label, extra bench args
  local label=$1; shift
  timeout 300 python bench.py --config 4 --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$label', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,3), 'M cells/s', round(r['frac'],4), d.get('parity_check'), {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items()})"
}
for rep in 1 2; do
  run "weight  " --analog-kind weight_analogs --parity-only
  run "regr    " --analog-estimator regression --parity-only
done
run "mean    "
