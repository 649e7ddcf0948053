#!/bin/bash
# A/B against the previous library (lib/libsd_downscale_prev.so, if present) and phase ablations of the fused BcsdTemperature
# kernel (development library, SD_FZ_ABLATE); run through gpurun; log -> gpurun_out/exp_fz.log
set -u
O=gpurun_out; mkdir -p $O
L=$PWD/scikit-downscale_amd/lib
LOG=$O/exp_fz.log
: > $LOG
one() {  # label, env assignments...
  local label=$1; shift
  local out
  out=$(env "$@" timeout 200 python bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 2>&1 | tail -1)
  python - "$label" "$out" <<'PY' >> $LOG
import sys, json
try:
    d = json.loads(sys.argv[2])
    print('%-34s %8.3f ms  frac %.4f  parity %s  %s' % (sys.argv[1], d['ms_per_step'], d['roofline']['frac'], d.get('parity_check'), d['roofline'].get('per_kernel_avg_ms')))
except Exception as e:
    print('%-34s FAILED %s' % (sys.argv[1], sys.argv[2][-300:]))
PY
  tail -1 $LOG
}
if [ "${TESTS:-1}" = 1 ]; then
  timeout 900 python -m pytest tests/test_gpu_bcsd.py tests/test_gpu_detrend.py -x -q -m gpu 2>&1 | tail -3 | tee -a $LOG
fi
for rep in 1 2; do
  [ -f $L/libsd_downscale_prev.so ] && one prev SD_DOWNSCALE_LIB=$L/libsd_downscale_prev.so
  one new SD_DOWNSCALE_LIB=$L/libsd_downscale.so
done
D=SD_DOWNSCALE_LIB=$L/libsd_downscale_dev.so
one dev $D
one "dev SLAB" $D SD_FZ_SLAB=1
for a in 8 4 12 32 64 1 2 3 16 127; do one "dev ABLATE=$a" $D SD_FZ_ABLATE=$a; done
one "dev RS_SPLIT=0" $D SD_RS_SPLIT=0
cat $LOG
