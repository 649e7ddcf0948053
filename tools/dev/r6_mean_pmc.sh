#!/bin/bash
# SQ counters of analog_f1_mean_kernel (weight_analogs, 16 384 cells): what the kernel is busy with
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/meanpmc; mkdir -p $O
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
ARGS="--config 4 --cells 16384 --steps 2 --warmup 1 --no-cpu-baseline --analog-kind weight_analogs --parity-only"
(cd /tmp && SD_ANALOG_NOBUCKETS=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p1 -o p -- python $R/bench.py $ARGS > $O/p1.log 2>&1)
(cd /tmp && SD_ANALOG_NOBUCKETS=1 timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/p2 -o p -- python $R/bench.py $ARGS > $O/p2.log 2>&1)
for p in p1 p2; do f=$(find $O/$p -name "*counter_collection.csv" | head -1); python tools/dev/pmc_summary.py $f mean_kernel; done
