for i in 1 2; do
for lib in libsd_downscale_prev.so libsd_downscale.so; do
SD_DOWNSCALE_LIB=/root/repo/scikit-downscale_amd/lib/$lib timeout 200 python bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['per_kernel_avg_ms'])"
done; done
