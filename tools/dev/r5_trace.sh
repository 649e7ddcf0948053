#!/bin/bash
# phase clocks of the fused kernel: tools/dev/r5_trace.sh [lib under scikit-downscale_amd/lib] (a -DSD_DEV build)
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/${1:-libsd_downscale_dev.so}
mkdir -p gpurun_out
SD_FZ_ABLATE=2048 SD_FX_TRACE=$PWD/gpurun_out/fx_trace.bin timeout 200 python bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print('$1 traced', d['ms_per_step'], d['roofline']['per_kernel_avg_ms'])"
python tools/dev/trace_fx.py gpurun_out/fx_trace.bin ${2:-0.000476}
