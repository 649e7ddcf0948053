#!/bin/bash
# round 6: the prefix-sum reads of the fused analog kernel's mean / spread phases issued kOutQ queries at a time
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
SD_FUSED_TRACE=1 python tools/dev/trace_fused.py 2>&1 | grep -v "^$" | tail -10
for rep in 1 2; do
timeout 300 python bench.py --config 4 --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('mean', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,3), 'M cells/s', round(r['frac'],4), {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if v*r['launches_per_step'][k] > 0.5})"
done
