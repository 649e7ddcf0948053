export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
run() {
  local label=$1; shift
  timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --parity-only "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$label', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,3), 'M cells/s', d.get('parity_check'), {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items()})"
}
run "weight mean-kernel runs " --analog-kind weight_analogs
SD_ANALOG_NOPREFIX=1 run "weight window-kernel runs" --analog-kind weight_analogs
SD_ANALOG_NOPREFIX=1 SD_ANALOG_NORUNS=1 run "weight window-kernel noruns" --analog-kind weight_analogs
