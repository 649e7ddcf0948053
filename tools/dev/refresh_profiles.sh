#!/bin/bash
# Refresh the round's measurement files (run on the GPU box through gpurun; outputs under gpurun_out/r, copied to profiles/rNN by hand).
set -u
O=gpurun_out/r; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
# Order matters: `pmc` needs a bench line per config for the kernel names / launches per step (it makes a short one itself when
# there is none), writes profiles/pmc_traffic.json, and the `bench` lines made afterwards in the same job carry that figure
# with `traffic_source.kernel_sources_match: true`.
PARTS=${PARTS:-pmc bench stats sq extra torchrun tests}
CONFIGS=${CONFIGS:-2 3 4}
for part in $PARTS; do
case $part in
bench)
  for c in $CONFIGS; do
    case $c in
      2) timeout 600 python bench.py > $O/bench_config2.json 2> $O/bench_config2.err;;
      3) timeout 300 python bench.py --config 3 --steps 60 > $O/bench_config3.json 2> $O/bench_config3.err;;
      4) timeout 300 python bench.py --config 4 --steps 40 > $O/bench_config4.json 2> $O/bench_config4.err;;
    esac
  done
  ;;
stats)
  for c in $CONFIGS; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o c$c -- python $R/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/rocprof_c$c.log 2>&1)
    find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_config$c.csv \;
    rm -rf $O/prof
  done
  ;;
pmc)
  for c in $CONFIGS; do
    [ -f $O/bench_config$c.json ] || timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err
    cells=""
    for ctr in FETCH_SIZE WRITE_SIZE; do
      n=$(echo $ctr | tr A-Z a-z | sed 's/_size//')
      (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$O/pmc -o p -- python $R/bench.py --config $c $cells --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/pmc_${n}_c$c.log 2>&1)
      find $O/pmc -name "*counter_collection.csv" -exec cp {} $O/pmc_${n}_c$c.csv \;
      rm -rf $O/pmc
    done
  done
  python tools/dev/pmc_traffic.py $O $O/pmc_traffic.json 2>&1 | tail -4
  cp $O/pmc_traffic.json profiles/pmc_traffic.json   # (on the GPU box: read by the bench lines of this job; copied back by refresh_all.sh)
  ;;
sq)
  for c in $CONFIGS; do
    [ $c = 3 ] && continue
    cells=""; [ $c = 4 ] && cells="--cells 16384"
    (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$O/pmc -o p -- python $R/bench.py --config $c $cells --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/pmc_sq1_c$c.log 2>&1)
    find $O/pmc -name "*counter_collection.csv" -exec cp {} $O/pmc_sq1_c$c.csv \;
    rm -rf $O/pmc
    (cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/$O/pmc -o p -- python $R/bench.py --config $c $cells --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/pmc_sq2_c$c.log 2>&1)
    find $O/pmc -name "*counter_collection.csv" -exec cp {} $O/pmc_sq2_c$c.csv \;
    rm -rf $O/pmc
  done
  ;;
extra)
  # (the analog kinds / AnalogRegression / F = 3 are part of the default bench line since round 5: secondary.analog_kinds, analog_f3)
  rm -f $O/bench_extra.json
  timeout 200 python tools/bench_extra.py --workload pure_regression --cells 100000 --out $O/bench_extra.json > /dev/null 2>&1
  for w in qmr ecm; do timeout 200 python tools/bench_extra.py --workload $w --cells 100000 --out $O/bench_extra.json > /dev/null 2>&1; done
  wc -l $O/bench_extra.json
  ;;
torchrun)
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_torchrun_1proc.json 2> $O/bench_torchrun_1proc.err
  tail -c 400 $O/bench_torchrun_1proc.json
  ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
  grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
  ;;
esac
done
for c in 2 3 4; do [ -f $O/bench_config$c.json ] && python - $O/bench_config$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['config']['baseline_config'], round(d['value']), d['ms_per_step'], round(d['roofline']['frac'],4), d.get('cpu_baseline',{}).get('value'), d.get('end_to_end'))
PY
done
