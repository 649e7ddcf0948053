#!/bin/bash
# round 6: value-ordered query runs (sd_analog_runs.h) against the time-ordered staging, same box, development library
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
run() {  # label, extra bench args
  local label=$1; shift
  timeout 300 python bench.py --config 4 --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$label', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,3), 'M cells/s', round(r['frac'],4), {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items()})"
}
for rep in 1 2; do
  run "mean runs   "
  SD_ANALOG_NORUNS=1 run "mean noruns "
done
run "weight runs  " --analog-kind weight_analogs --parity-only
SD_ANALOG_NORUNS=1 run "weight noruns" --analog-kind weight_analogs --parity-only
run "regr runs    " --analog-estimator regression --parity-only
SD_ANALOG_NORUNS=1 run "regr noruns  " --analog-estimator regression --parity-only
run "best runs    " --analog-kind best_analog --analog-k 200 --parity-only
SD_ANALOG_NORUNS=1 run "best noruns  " --analog-kind best_analog --analog-k 200 --parity-only
