set -u
O=gpurun_out/r4x; mkdir -p $O
timeout 120 scikit-downscale_amd/csrc/microbench/wsort_test 65536 16 > $O/microbench_wsort.log 2>&1; tail -3 $O/microbench_wsort.log
timeout 200 scikit-downscale_amd/csrc/microbench/wrank_test 32768 16 > $O/microbench_wrank.log 2>&1; tail -2 $O/microbench_wrank.log
timeout 300 python -m pytest tests/test_gpu_bcsd.py -x -q -m gpu -k "chunk or lazy or blocks" 2>&1 | tail -2
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --cells 20000 --steps 10 --warmup 2 > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err; python -c "
import json;d=json.loads(open('$O/bench_2ranks_1gpu.json').read().strip().splitlines()[-1]);print(d['n_gpus'], round(d['value']), d.get('value_with_gather'), d.get('error'), str(d.get('scaling_claim'))[:80], d.get('value_with_host_gather'))"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_torchrun_1proc.json 2> $O/bench_torchrun_1proc.err; tail -c 300 $O/bench_torchrun_1proc.json
python - <<'PY'
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import bench
from skdownscale_amd import synth
index = synth.daily_calendar(14600)
r = bench.pointwise_end_to_end(index, 0, 100000)
print("pointwise standalone", round(r["value"]), r["seconds"], r["best_seconds"])
PY
