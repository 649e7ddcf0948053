#!/bin/bash
# round 6: queries a thread searches together (SD_SEARCHQ = 2 / 4 / 8; variant libraries from tools/dev/build_variant.sh), config 4 and best_analog
mkdir -p gpurun_out/r6
run() {
  local label=$1; shift
  timeout 300 python bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --parity-only "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$label', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,3), 'M cells/s', d.get('parity_check'), {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if v*r['launches_per_step'][k] > 0.5})"
}
for rep in 1 2; do
  run "mean  searchq=4"
  SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_v_sq2.so run "mean  searchq=2"
  SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_v_sq8.so run "mean  searchq=8"
done
run "best  searchq=4" --analog-kind best_analog --analog-k 200
SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_v_sq2.so run "best  searchq=2" --analog-kind best_analog --analog-k 200
SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_v_sq8.so run "best  searchq=8" --analog-kind best_analog --analog-k 200
run "weight searchq=4(mean kernel: SD_MEANQ)" --analog-kind weight_analogs
