#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_bcsd.py tests/test_gpu_qm.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/pytest_bcsd.log
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
b slab SD_DOWNSCALE_LIB=$DEV SD_FZ_SLAB=1
b search SD_DOWNSCALE_LIB=$DEV SD_BCSD_FUSED=0
b abl3 SD_DOWNSCALE_LIB=$DEV SD_FZ_ABLATE=3
b abl100 SD_DOWNSCALE_LIB=$DEV SD_FZ_ABLATE=100
b prod2 A=1
pass() {  # name, counters...
    local name=$1; shift
    (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o $name -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_$name.log 2>&1)
    find gpurun_out/pmc_$name -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_$name.csv \;
    rm -rf gpurun_out/pmc_$name
    python tools/dev/pmc_summary.py gpurun_out/pmc_$name.csv bcsd_fz 2>&1 | cut -c1-600
}
pass sq1 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
