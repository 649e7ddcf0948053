#!/bin/bash
# A/B of whole libraries: exp_ab.sh lib1.so lib2.so ... (two rounds, log -> gpurun_out/exp_ab.log)
set -u
O=gpurun_out; mkdir -p $O
L=$PWD/scikit-downscale_amd/lib
LOG=$O/exp_ab.log; : > $LOG
for rep in 1 2; do
for lib in "$@"; do
  out=$(SD_DOWNSCALE_LIB=$L/$lib timeout 200 python bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 ${BENCH_ARGS:-} 2>&1 | tail -1)
  python - "$lib" "$out" <<'PY' | tee -a $LOG
import sys, json
try:
    d = json.loads(sys.argv[2])
    print('%-28s %8.3f ms  frac %.4f  %s' % (sys.argv[1], d['ms_per_step'], d['roofline']['frac'], {k: round(v, 3) for k, v in d['roofline'].get('per_kernel_avg_ms').items()}))
except Exception as e:
    print('%-28s FAILED %s' % (sys.argv[1], sys.argv[2][-300:]))
PY
done; done
