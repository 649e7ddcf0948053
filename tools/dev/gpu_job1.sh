#!/bin/bash
# GPU job: full GPU test suite, then the headline bench on the production library and on development variants.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest.log
tail -3 gpurun_out/pytest.log
DEV=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
b() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/bench_{n}.json').read().strip().split('\n')[-1])
    print(n, 'ms/step %.3f'%d['ms_per_step'], 'kernel_ms %.3f'%d['roofline']['kernel_ms_per_step'], 'frac %.4f'%d['roofline']['frac'], d['roofline']['per_kernel_avg_ms'], d.get('parity_check'))
except Exception as e:
    print(n, 'FAILED', e, open(f'gpurun_out/bench_{n}.err').read()[-600:])
PY
}
b prod A=1
b dev_slab SD_DOWNSCALE_LIB=$DEV SD_FZ_SLAB=1
b dev_search SD_DOWNSCALE_LIB=$DEV SD_BCSD_FUSED=0
b prod2 A=1
