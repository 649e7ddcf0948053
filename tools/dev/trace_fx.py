#!/usr/bin/env python3
"""Phase clocks of the fused BcsdTemperature kernel (development library, SD_FZ_ABLATE=2048 SD_FX_TRACE=<file>):
mean / median time per phase over the sampled workgroups (wave 0 of each), in microseconds (s_memtime at 100 MHz)."""
import sys
import numpy as np

NAMES = ["x tiles requested + column sums", "x_fut tile committed + barrier", "rolling mean", "u stored + keys", "sort of u", "fix-up of u",
         "vote barrier", "y tile requested", "y tile committed + barrier", "y_climo + keys", "sort of y", "fix-up of y + gather",
         "map + scatter + shift", "last barrier", "stores issued"]
if len(sys.argv) > 3 and sys.argv[3] == "fd":  # bcsd_fd_kernel (round 6: tiles by LDS-DMA)
    NAMES = ["x_hist summed + x_fut requested", "x_fut landed + barrier", "rolling mean", "keys + u2 + window barrier + early y requests", "sort of u",
             "fix-up of u", "vote barrier", "late y requested", "y landed + barrier", "y_climo + keys", "sort of y", "fix-up of y + gather",
             "scatter + shift", "last barrier", "stores issued"]
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 16)
a = a[(a[:, 0] != 0) & (a[:, 15] > a[:, 0])]
tick_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
d = np.diff(a, axis=1) * tick_us
print(f"{len(a)} workgroups sampled; lifetime mean {d.sum(1).mean():.2f} us, median {np.median(d.sum(1)):.2f} us")
for i, nme in enumerate(NAMES):
    print(f"  {nme:34s} mean {d[:, i].mean():7.2f}  median {np.median(d[:, i]):7.2f}  p90 {np.percentile(d[:, i], 90):7.2f}")
