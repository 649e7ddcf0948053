#!/bin/bash
# development library: phase clocks of analog_f1_mean3_kernel (block 0, first cells); then tests + bench on the production library
mkdir -p gpurun_out
SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so SD_M3_TRACE=1 timeout 300 python bench.py --config 4 --cells 16384 --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/m3_trace.json 2> gpurun_out/m3_trace.err
grep "mean3 trace" gpurun_out/m3_trace.err | tail -4
timeout 900 python -m pytest tests/test_gpu_analog.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_analog.log
tail -3 gpurun_out/pytest_analog.log
timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c4.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_kernel_avg_ms'])
PY
