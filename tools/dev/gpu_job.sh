export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
for nt in 8 16 24; do echo threads $nt; SD_COPY_THREADS=$nt timeout 300 python tools/dev/host_api_rate.py 2>&1 | grep "bcsd_predict (1\|fit + predict"; done
echo no stream; SD_COPY_NOSTREAM=1 timeout 300 python tools/dev/host_api_rate.py 2>&1 | grep "bcsd_predict (1\|fit + predict"
SD_COPY_TRACE=1 timeout 300 python tools/dev/host_api_rate.py 2>&1 | grep chunk | tail -8
