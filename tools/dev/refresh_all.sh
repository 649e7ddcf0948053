#!/bin/bash
# Regenerate the round's measurement files in ONE gpurun job and file them under profiles/ (run here, in the build container):
#   tools/dev/refresh_all.sh r03            # PARTS / CONFIGS as for refresh_profiles.sh
# The PMC passes run first and rewrite profiles/pmc_traffic.json (with the HEAD and a hash of the kernel sources), so the bench
# lines of the same job carry a traffic figure that belongs to the code they time.
set -eu
R=${1:?round directory name, e.g. r03}
HEAD=$(git rev-parse --short HEAD)
rm -rf gpurun_out/r
/usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-2400} -- "SD_PROFILE_HEAD=$HEAD PARTS='${PARTS:-pmc bench stats sq tests}' CONFIGS='${CONFIGS:-2 3 4}' bash tools/dev/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; tail -20 gpurun_out/refresh.log"
mkdir -p profiles/$R
cp gpurun_out/r/*.json gpurun_out/r/*.csv profiles/$R/ 2>/dev/null || true
[ -f gpurun_out/r/pytest_gpu.log ] && cp gpurun_out/r/pytest_gpu.log profiles/$R/
[ -f gpurun_out/r/pmc_traffic.json ] && cp gpurun_out/r/pmc_traffic.json profiles/pmc_traffic.json
ls profiles/$R | head -40
