#!/bin/bash
# Refresh the round's measurement files (run on the GPU box through gpurun; outputs under gpurun_out/r).
set -u
O=gpurun_out/r; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py > $O/bench_100k.json 2> $O/bench_100k.err
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/rocprof_bench.log 2>&1)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_bench.csv \;
rm -f $O/bench_extra.json
timeout 200 python tools/bench_extra.py --workload bcsd_pr --cells 250000 --out $O/bench_extra.json > /dev/null 2>&1
for kind in mean_analogs best_analog weight_analogs; do timeout 200 python tools/bench_extra.py --workload analog --kind $kind --cells 16384 --out $O/bench_extra.json > /dev/null 2>&1; done
timeout 200 python tools/bench_extra.py --workload analog --cells 100000 --out $O/bench_extra.json > /dev/null 2>&1
timeout 200 python tools/bench_extra.py --workload analogreg --cells 16384 --out $O/bench_extra.json > /dev/null 2>&1
for f in 2 3 4; do timeout 200 python tools/bench_extra.py --workload analog --features $f --cells 2048 --out $O/bench_extra.json > /dev/null 2>&1; done
timeout 200 python tools/bench_extra.py --workload analogreg --features 3 --cells 2048 --out $O/bench_extra.json > /dev/null 2>&1
timeout 200 python tools/bench_extra.py --workload pure_regression --cells 100000 --out $O/bench_extra.json > /dev/null 2>&1
for w in qmr ecm; do timeout 200 python tools/bench_extra.py --workload $w --cells 100000 --out $O/bench_extra.json > /dev/null 2>&1; done
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_torchrun_1.json 2> $O/bench_torchrun_1.err
rm -rf $O/prof
tail -c 600 $O/bench_100k.json; wc -l $O/bench_extra.json; tail -c 300 $O/bench_torchrun_1.json; head -5 $O/rocprofv3_kernel_stats_bench.csv
