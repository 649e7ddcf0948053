run() {  # label, lib
  local label=$1; shift
  SD_ANALOG_NOBUCKETS=1 SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/$1 timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --analog-kind weight_analogs --parity-only 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('$label', round(d['ms_per_step'],2), 'ms', {k: round(v*r['launches_per_step'][k],2) for k,v in r['per_kernel_avg_ms'].items() if 'mean' in k})"
}
run "runs full               " libsd_downscale_dev.so
run "runs, no window loads   " libsd_v_m_nowin.so
run "runs, no window, no xs  " libsd_v_m_noxs.so
run "runs, no rcp / Newton   " libsd_v_m_norcp.so
run "runs, no search         " libsd_v_m_nosearch.so
run "runs, none of them      " libsd_v_m_nothing.so
