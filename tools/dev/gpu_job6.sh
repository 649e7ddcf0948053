#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/pytest.log
tail -4 gpurun_out/pytest.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 3000 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err
tail -c 600 gpurun_out/bench_torchrun1.json; tail -3 gpurun_out/bench_torchrun1.err
