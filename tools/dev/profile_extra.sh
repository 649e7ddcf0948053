#!/bin/bash
# rocprofv3 kernel statistics of the secondary BASELINE configs (run through gpurun; outputs under gpurun_out/r).
set -u
O=gpurun_out/r; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
run() {  # name, bench_extra arguments...
    local name=$1; shift
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$name -o $name -- python $R/tools/bench_extra.py "$@" > $R/$O/rocprof_$name.log 2>&1)
    find $O/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_$name.csv \;
    rm -rf $O/prof_$name
    tail -1 $O/rocprof_$name.log | cut -c1-200
    head -6 $O/rocprofv3_kernel_stats_$name.csv
}
run bcsd_pr_250k --workload bcsd_pr --cells 250000 --steps 5
run pureanalog_100k --workload analog --cells 100000 --steps 5
run pureanalog_f3 --workload analog --features 3 --cells 2048 --steps 3
