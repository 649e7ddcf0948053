#!/bin/bash
# A/B variants of one translation unit: tools/dev/build_variant.sh <name> <unit (e.g. sd_bcsd_fx)> "<-D flags>"
# -> scikit-downscale_amd/lib/libsd_v_<name>.so (the other units are the production objects; run `make` first)
set -eu
name=$1; unit=$2; flags=${3:-}
cd "$(dirname "$0")/../../scikit-downscale_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result $flags -c csrc/$unit.hip -o csrc/$unit.v_$name.o 2>&1 | grep -v "hip-link" || true
objs=""
for u in sd_ctx sd_bcsd sd_bcsd_rs sd_bcsd_fx sd_analog sd_qm sd_linreg sd_comm; do
  if [ $u = $unit ]; then objs="$objs csrc/$u.v_$name.o"; else objs="$objs csrc/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o lib/libsd_v_$name.so -ldl -Wl,-rpath,/opt/rocm/lib
ls -la lib/libsd_v_$name.so
