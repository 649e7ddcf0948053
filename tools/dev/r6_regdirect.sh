#!/bin/bash
# round 6: AnalogRegression k = 30 with the window summed directly (reg_batch, no prefix arrays) against the prefix differences
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
run() {  # label, extra bench args
  local label=$1; shift
  timeout 300 python bench.py --config 4 --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$label', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,3), 'M cells/s', round(r['frac'],4), d.get('parity_check'), {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if v*r['launches_per_step'][k] > 0.5})"
}
for rep in 1 2; do
  run "regr direct          " --analog-estimator regression --parity-only
  SD_ANALOG_QSPLIT=2 run "regr direct qsplit=2 " --analog-estimator regression --parity-only
  SD_ANALOG_REG_PREFIX=1 run "regr prefix          " --analog-estimator regression --parity-only
done
SD_ANALOG_NORUNS=1 run "regr direct, no runs " --analog-estimator regression --parity-only
