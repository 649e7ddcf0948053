#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_analog.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_analog.log
tail -3 gpurun_out/pytest_analog.log
timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err
tail -c 1500 gpurun_out/bench_c4.json; tail -3 gpurun_out/bench_c4.err
