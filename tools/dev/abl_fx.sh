#!/bin/bash
# ablation sweep of the fused kernel on the development library (SD_FZ_ABLATE bits, see csrc/sd_bcsd_fx.hip): tools/dev/abl_fx.sh 0 32 64 ...
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
for a in "$@"; do
  SD_FZ_ABLATE=$a timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('abl=$a', {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if 'fx' in k})"
done
