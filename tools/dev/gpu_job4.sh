#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_bcsd.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/pytest_bcsd.log
tail -3 gpurun_out/pytest_bcsd.log
DEV=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
b() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/bench_{n}.json').read().strip().split('\n')[-1])
    print(n, 'ms/step %.3f'%d['ms_per_step'], 'kernel_ms %.3f'%d['roofline']['kernel_ms_per_step'], {k:round(v,3) for k,v in d['roofline']['per_kernel_avg_ms'].items()})
except Exception as e:
    print(n, 'FAILED', e, open(f'gpurun_out/bench_{n}.err').read()[-600:])
PY
}
b prod A=1
b nopf SD_DOWNSCALE_LIB=$DEV SD_FZ_ABLATE=128
for d in 2 4 6 8 10 12 16; do b succ$d SD_DOWNSCALE_LIB=$DEV SD_FZ_ABLATE=$((d*1048576)); done
b slab SD_DOWNSCALE_LIB=$DEV SD_FZ_SLAB=1
b slab_succ8 SD_DOWNSCALE_LIB=$DEV SD_FZ_SLAB=1 SD_FZ_ABLATE=$((8*1048576))
b prod2 A=1
