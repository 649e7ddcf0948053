#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_bcsd.py tests/test_gpu_qm.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -60 > gpurun_out/pytest_bcsd.log
tail -4 gpurun_out/pytest_bcsd.log
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
for t in 10 20 35 50 70 100 150; do b stag$t SD_DOWNSCALE_LIB=$DEV SD_FZ_ABLATE=$((t*256)); done
b slab SD_DOWNSCALE_LIB=$DEV SD_FZ_SLAB=1
b slab_stag50 SD_DOWNSCALE_LIB=$DEV SD_FZ_SLAB=1 SD_FZ_ABLATE=$((50*256))
