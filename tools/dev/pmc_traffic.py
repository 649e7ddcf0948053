"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/dev/refresh_profiles.sh.

usage: pmc_traffic.py <dir with pmc_fetch_c<N>.csv / pmc_write_c<N>.csv / bench_config<N>.json> <out.json>

Per config: HBM bytes per timed step = sum over the kernels bench.py times of (mean counter value per dispatch x launches
per step).  Counter unit KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests
at 64 bytes -> x2 (calibrated here on a known byte count: profiles/r01/pmc_calibration_fetch_size_tile_read.csv);
WRITE_SIZE is exact on the same calibration.
"""
import collections
import csv
import json
import os
import sys

FETCH_CORRECTION = 2.0


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    base = n.split("<")[0].split("::")[-1]
    # the whole-lane instantiations of the fused BCSD kernels (last template argument FULL = true) are launched -- and named
    # by SD_LAUNCH -- separately from the general ones
    if base in ("bcsd_fx_kernel", "bcsd_fxp_kernel", "bcsd_fxc_kernel") and "<" in n and n.rstrip().rstrip(">").rstrip().endswith("true"):
        base += "_full"
    return base


def per_kernel(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}  # KiB -> bytes, mean per dispatch


def main():
    src, out = sys.argv[1], sys.argv[2]
    entries = []
    for config in (2, 3, 4):
        bj = os.path.join(src, f"bench_config{config}.json")
        fj, wj = os.path.join(src, f"pmc_fetch_c{config}.csv"), os.path.join(src, f"pmc_write_c{config}.csv")
        if not (os.path.exists(bj) and os.path.exists(fj) and os.path.exists(wj)):
            continue
        line = json.loads(open(bj).read().strip().splitlines()[-1])
        roof = line["roofline"]
        fetch, write = per_kernel(fj), per_kernel(wj)
        total, detail = 0.0, {}
        for kname, launches in roof["launches_per_step"].items():
            # bench.py names (SD_LAUNCH) are prefixes / variants of the symbol names: bcsd_rs_rank_kernel = bcsd_rs_kernel<K, 3, ...>
            sym = {"bcsd_rs_rank_kernel": "bcsd_rs_kernel", "bcsd_rs_apply_kernel": "bcsd_rs_kernel", "bcsd_rs_fit_kernel": "bcsd_rs_kernel",
                   "analog_sort2_exact_kernel": "analog_sort2_kernel", "bcsd_fxp_kernel_list": "bcsd_fxp_kernel"}.get(kname, kname)
            f, w = fetch.get(sym, 0.0) * FETCH_CORRECTION, write.get(sym, 0.0)
            detail[kname] = {"symbol": sym, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "launches_per_step": launches}
            total += (f + w) * launches
        entries.append({
            "workload": {"config": config, "cells": line["config"]["cells_per_gpu"], "timesteps": line["config"]["timesteps"], "kernel": roof["kernel"]},
            "traffic_bytes_per_step": total,
            "algorithmic_bytes_per_step": roof["algorithmic_bytes_per_step"],
            "ratio": total / roof["algorithmic_bytes_per_step"],
            "per_kernel": detail,
        })
    import hashlib
    import subprocess
    import time

    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lib = os.path.join(root, "scikit-downscale_amd", "lib", "libsd_downscale.so")
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:  # noqa: BLE001
        head = None
    head = os.environ.get("SD_PROFILE_HEAD", head)  # (the GPU box has no .git: refresh_profiles.sh passes the HEAD along)
    doc = {
        "source": {"generated": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "head": head, "dir": src,
                   "library_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None,
                   "kernel_sources_sha16": hashlib.sha256(b"".join(open(os.path.join(root, "scikit-downscale_amd", "csrc", f), "rb").read()
                                                                   for f in sorted(os.listdir(os.path.join(root, "scikit-downscale_amd", "csrc")))
                                                                   if f.endswith((".hip", ".h")))).hexdigest()[:16]},
        "_comment": "HBM traffic per timed step from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py "
                    "--config N --steps 2 --warmup 1 --no-cpu-baseline` (profiles/r02/pmc_fetch_cN.csv, pmc_write_cN.csv; counter unit KiB, "
                    "mean over the dispatches of a kernel symbol x launches per step).  gfx950: FETCH_SIZE x 2 (128-byte requests tallied at 64 "
                    "bytes, calibrated on a known byte count in profiles/r01); WRITE_SIZE exact.  Kernels that share a symbol (the RANK / APPLY "
                    "modes of bcsd_rs_kernel) share its mean.",
        "fetch_correction_factor": FETCH_CORRECTION,
        "entries": entries,
    }
    json.dump(doc, open(out, "w"), indent=1)
    for e in entries:
        print(e["workload"]["config"], "%.1f GB per step, %.2fx the algorithmic bytes" % (e["traffic_bytes_per_step"] / 1e9, e["ratio"]))


if __name__ == "__main__":
    main()
