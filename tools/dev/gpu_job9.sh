#!/bin/bash
mkdir -p gpurun_out
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
for v in 0 1; do
  if [ $v = 1 ]; then export SD_M3_NOTOUCH=1; fi
  timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench_c4_t$v.json 2> gpurun_out/bench_c4_t$v.err
  python - gpurun_out/bench_c4_t$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_kernel_avg_ms'])
PY
done
