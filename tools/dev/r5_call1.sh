#!/bin/bash
# round 5, GPU call 1: FULL-lane fused BcsdTemperature kernel -- parity tests, A/B against the round-4 library, instruction counters
set -u
O=gpurun_out/c1; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
L=$R/scikit-downscale_amd/lib
ab() {  # name lib config steps
  SD_DOWNSCALE_LIB=$2 timeout 300 python bench.py --config $3 --steps $4 --warmup 3 --parity-only 2> $O/ab_$1_c$3.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1 config $3', d['ms_per_step'], 'ms/step frac', round(d['roofline']['frac'],4), 'kernels', {k: round(v,3) for k,v in d['roofline'].get('per_kernel_avg_ms',{}).items()}, 'parity', d.get('parity_check'))"
}
timeout 900 python -m pytest tests/test_gpu_bcsd.py tests/test_gpu_fuzz.py tests/test_gpu_detrend.py -m gpu -x -q > $O/pytest_bcsd.log 2>&1; tail -5 $O/pytest_bcsd.log
for i in 1 2; do
  ab r4 $L/libsd_r4.so 2 30
  ab new $L/libsd_downscale.so 2 30
done
SD_FX_NOFULL=1 ab new_nofull $L/libsd_downscale_dev.so 2 30
ab r4 $L/libsd_r4.so 3 15
ab new $L/libsd_downscale.so 3 15
# instruction counters of the new kernel
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/$O/pmc -o p -- python $R/bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/pmc_sq.log 2>&1)
find $O/pmc -name "*counter_collection.csv" -exec cp {} $O/pmc_sq_c2.csv \;
rm -rf $O/pmc
python tools/dev/pmc_summary.py $O/pmc_sq_c2.csv bcsd_fx 2>&1 | tail -12
