export SD_DOWNSCALE_LIB=$PWD/scikit-downscale_amd/lib/libsd_downscale_dev.so
timeout 900 python tools/dev/fuzz_gpu.py 300 701 2>&1 | tail -1
timeout 600 python tools/dev/fuzz_fx.py 400 702 2>&1 | tail -1
timeout 600 python tools/dev/fuzz_fd.py 300 703 2>&1 | tail -1
timeout 600 python tools/dev/fuzz_runs.py 150 704 2>&1 | tail -1
timeout 600 python tools/dev/fuzz_analog_fused.py 400 705 2>&1 | tail -1
timeout 600 python tools/dev/fuzz_topk.py 400 706 2>&1 | tail -1
timeout 600 python tools/dev/fuzz_pointwise.py 120 2>&1 | tail -1
timeout 900 python tools/dev/fuzz_mean_kernel.py 200 707 2>&1 | tail -1
