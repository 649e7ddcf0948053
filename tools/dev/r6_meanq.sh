#!/bin/bash
# queries a thread of analog_f1_mean_kernel searches together (SD_MEANQ) after the straight-line batches
run() {  # label, lib, args
  local label=$1; local lib=$2; shift; shift
  SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/$lib timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --parity-only "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$label', round(d['ms_per_step'],2), 'ms', {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if 'mean' in k})"
}
for lib in libsd_downscale_dev.so libsd_v_mq1.so libsd_v_mq4.so; do
  run "weight $lib" $lib --analog-kind weight_analogs
  run "regr   $lib" $lib --analog-estimator regression
done
