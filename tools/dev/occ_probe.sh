#!/bin/bash
# occupancy probe: the fused kernel at different series lengths (LDS rows shrink with the month length)
for t in 14600 14400 14000 13000; do
  timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --times $t 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print('times=$t', round(d['roofline']['per_kernel_avg_ms']['bcsd_fx_kernel'],2),'ms', round(d['roofline']['frac'],3))"
done
