#!/bin/bash
# PMC passes for the PureAnalog pipeline (16 384 cells x 14 600, k=30): LDS counters, HBM fetch and write sizes.
set -u
O=gpurun_out/r; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
pass() {  # name, counters...
    local name=$1; shift
    (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$O/pmc_$name -o $name -- python $R/tools/bench_extra.py --workload analog --cells 16384 --steps 2 > $R/$O/pmc_$name.log 2>&1)
    find $O/pmc_$name -name "*counter_collection.csv" -exec cp {} $O/pmc_analog_$name.csv \;
    rm -rf $O/pmc_$name
    ls -la $O/pmc_analog_$name.csv
}
pass lds SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
for n in lds fetch write; do python tools/dev/pmc_summary.py $O/pmc_analog_$n.csv analog 2>&1 | cut -c1-400; done
