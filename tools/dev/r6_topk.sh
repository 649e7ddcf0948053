#!/bin/bash
# round 6: analog_slab_topk_kernel (F = 3, k = 30): counters, phase clocks and the prune threshold (development library)
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
run() {
  python bench.py --config 4 --analog-features 3 --cells ${CELLS:-8192} --steps 1 --warmup 1 --no-cpu-baseline --parity-only 2>/tmp/err.txt | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$1', round(d['ms_per_step'],2), 'ms', round(d['value']), 'cells/s', d.get('parity_check'), {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if 'slab_' in k})"
  grep -E "^\[topk\]|^\[slab\]" /tmp/err.txt | tail -3
}
for pa in ${PRUNE_ATS:-16}; do
  SD_TOPK_READLANE=1 SD_TOPK_PRUNE_AT=$pa run "readlane filter, prune_at=$pa"
  SD_TOPK_PRUNE_AT=$pa run "prune_at=$pa"
  SD_TOPK_PRUNE_AT=$pa SD_ANALOG_ABLATE=4 CELLS=2048 run "prune_at=$pa (counters)"
done
SD_ANALOG_HEAP=1 run "heap kernel"
