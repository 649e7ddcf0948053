mkdir -p gpurun_out
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --cells 20000 --steps 10 --warmup 2 > gpurun_out/bench_2ranks_1gpu.json 2> gpurun_out/bench_2ranks_1gpu.err
tail -c 1800 gpurun_out/bench_2ranks_1gpu.json; echo; tail -5 gpurun_out/bench_2ranks_1gpu.err
