#!/bin/bash
# round 6: bcsd_fd_kernel (tiles by LDS-DMA) against bcsd_fx_kernel<20, true, true> on the same box, development library
export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
mkdir -p gpurun_out/r6
run() {
  timeout 200 python bench.py --steps ${STEPS:-15} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$1', round(d['ms_per_step'],3), round(r['frac'],4), {k: round(v*r['launches_per_step'][k],3) for k,v in r['per_kernel_avg_ms'].items() if 'rs_' not in k})"
}
for rep in 1 2 3; do
  run "dma   "
  SD_FD_LATE=1 run "dmalate"
  SD_FX_NODMA=1 run "nodma "
done
for a in "$@"; do SD_FZ_ABLATE=$a run "dma abl=$a"; done
SD_FZ_ABLATE=2048 SD_FX_TRACE=gpurun_out/r6/trace_fd.bin STEPS=3 run "trace dma" 
python tools/dev/trace_fx.py gpurun_out/r6/trace_fd.bin 0.01 fd
SD_FD_LATE=1 SD_FZ_ABLATE=2048 SD_FX_TRACE=gpurun_out/r6/trace_fdl.bin STEPS=3 run "trace dmalate"
python tools/dev/trace_fx.py gpurun_out/r6/trace_fdl.bin 0.01 fd
SD_FX_NODMA=1 SD_FZ_ABLATE=2048 SD_FX_TRACE=gpurun_out/r6/trace_fx.bin STEPS=3 run "trace nodma"
python tools/dev/trace_fx.py gpurun_out/r6/trace_fx.bin 0.01
