#!/usr/bin/env python
"""fuzz_gpu.main over fresh seeds until the time is used: fuzz_loop.py [seconds] [first seed] [cases per seed]"""
import sys
import time
import traceback
import warnings

import fuzz_gpu

if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    t0, n = time.time(), 0
    warnings.simplefilter("ignore")
    while time.time() - t0 < seconds:
        try:
            fuzz_gpu.main(per, seed)
        except Exception:  # noqa: BLE001
            print(f"FAILED seed={seed} (cases per seed {per})", flush=True)
            traceback.print_exc(limit=3)
            sys.exit(1)
        n += per
        seed += 1
    print(f"fuzz_loop: {n} cases, seeds up to {seed - 1}, {time.time() - t0:.0f} s: all ok")
