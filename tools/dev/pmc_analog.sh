#!/bin/bash
# PMC passes for the PureAnalog pipeline (BASELINE config 4 at 16 384 cells x 14 600, k=30): SQ counters, HBM fetch and write sizes.
set -u
O=gpurun_out/r; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
pass() {  # name, counters...
    local name=$1; shift
    (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$O/pmc_$name -o $name -- python $R/bench.py --config 4 --cells 16384 --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/pmc_$name.log 2>&1)
    find $O/pmc_$name -name "*counter_collection.csv" -exec cp {} $O/pmc_analog_$name.csv \;
    rm -rf $O/pmc_$name
    ls -la $O/pmc_analog_$name.csv
}
for p in ${PASSES:-sq1 sq2 fetch write}; do
  case $p in
    sq1) pass sq1 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY;;
    sq2) pass sq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS;;
    fetch) pass fetch FETCH_SIZE;;
    write) pass write WRITE_SIZE;;
  esac
  python tools/dev/pmc_summary.py $O/pmc_analog_$p.csv analog 2>&1 | cut -c1-500
done
