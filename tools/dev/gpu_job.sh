mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_analog.py tests/test_gpu_fuzz.py tests/test_gpu_qm.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_analog.log
grep -n "passed\|failed" gpurun_out/pytest_analog.log
timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c4.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_kernel_avg_ms'])
PY
