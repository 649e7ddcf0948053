#!/usr/bin/env python
"""Headline benchmark: grid cells downscaled per second (fit + predict), 40-year daily series.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): BcsdTemperature quantile mapping, 100 000 cells x 14 600 daily
steps per GPU, float64, synthetic fields generated *in HBM* by the engine's counter-based generator
(mirror: skdownscale_amd/synth.py).  One step = one full pass of the hot path over the batch: fit
on (X_hist, y_obs) + predict on X_fut for every cell.  Inputs are HBM-resident before the timed
region.  With N > 1 ranks the cells shard across GPUs (weak scaling: fixed cells per GPU) and each
output shards stay resident on their GPU like the chunks of a dask-backed result; the RCCL gather of the
whole predicted field to rank 0 over xGMI is timed once outside the timed region (`gather_to_root_ms`)
or inside every step with --gather (root ingest of 7 x 11.7 GB per step is then the bound).

At N = 1 the product path is pure ctypes -> C ABI -> HIP (no torch import).  torch.distributed is
used only as launcher plumbing for N > 1 (barrier, max-over-ranks, RCCL gather).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "scikit-downscale_amd"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_CELL_STEP = 32  # X_hist + y_obs + X_fut read, out written: 4 x 8 B per (cell, time step)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=100_000, help="cells per GPU")
    ap.add_argument("--times", type=int, default=14_600)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fused", type=int, default=1, help="1: fused fit+predict entry point, 0: separate fit / predict")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: include the RCCL gather of the whole predicted field to rank 0 in every timed step "
                         "(default: shards stay resident on their GPU; the gather is timed once, outside the timed region)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-cells", type=int, default=32, help="cells verified against the oracle outside the timed region")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed plumbing even at N=1")
    return ap.parse_args()


def cpu_baseline(index, seed, c_full, target_seconds, check=None):
    """The plain-C restatement (oracle/sd_oracle.c, 'port') timed on this host's cores, bounded sample.

    The first (small) oracle run doubles as the checker of the engine's output: ``check(exp)`` receives the oracle's
    result for the first cells and returns a parity verdict (outside the timed region)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    from skdownscale_amd import synth

    if not c_oracle.available():
        return None, None
    threads = min(c_oracle.max_threads(), os.cpu_count() or 1)
    gid = (np.asarray(index.month) - 1).astype(np.int32)

    def run(n, seconds=0.0):
        """one pass over n cells, repeated until `seconds` of oracle time have been spent (same sample: it only has to
        be generated once, by the NumPy mirror of the device generator)"""
        cells = np.arange(n)
        X = synth.tas_field("X_hist", seed, index, cells, c_full)
        y = synth.tas_field("y_obs", seed, index, cells, c_full)
        Xp = synth.tas_field("X_fut", seed, index, cells, c_full)
        spent, reps = 0.0, 0
        while reps == 0 or (spent < seconds and reps < 16):
            t0 = time.perf_counter()
            out, st = c_oracle.bcsd_fit_predict(0, X, y, Xp, gid, gid, nthreads=threads)
            spent += time.perf_counter() - t0
            reps += 1
        return spent, reps, out

    n0 = 4 * threads
    dt, _, exp = run(n0)
    parity = check(exp) if check is not None else None
    rate = n0 / dt
    n = int(max(n0, min(rate * target_seconds, 8192)))
    dt, reps, _ = run(n, target_seconds)
    return {"value": n * reps / dt, "unit": "cells/s", "cores": threads, "kind": "port",
            "sample": f"{n} cells x {len(index)} steps x {reps} passes, oracle/sd_oracle.c (OpenMP, {threads} threads), "
                      f"{dt:.1f} s"}, parity


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:  # started by hand: re-launch one process per GPU
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
                                   os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    dist = torch = None
    if use_dist:
        import torch  # noqa: F811  (first, so the engine binds to the same HIP runtime as torch/RCCL)
        import torch.distributed as dist  # noqa: F811

        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from skdownscale_amd import _lib, synth
    from skdownscale_amd.engine import Context

    ctx = Context(local_rank)
    info = ctx.device_info()
    T, C = args.times, args.cells
    c_full = C * world
    c_off = C * rank
    index = synth.daily_calendar(T)
    gid = (np.asarray(index.month) - 1).astype(np.int32)
    tabs = synth.tas_tables(index)

    fields = {}
    for name in ("X_hist", "y_obs", "X_fut"):
        d = ctx.empty((T, C))
        tab = tabs[name]
        ctx.synth_fill(d, synth.GAUSS, args.seed, tab["stream"], c_offset=c_off, c_full=c_full, base=tab["base"],
                       amp=tab["amp"], cell_scale=tab["cell_scale"])
        fields[name] = d
    if use_dist:
        out_t = torch.empty((T, C), dtype=torch.float64, device=f"cuda:{local_rank}")
        out = ctx.wrap(out_t.data_ptr(), (T, C))
        gather_list = None
        if rank == 0:
            gather_list = [torch.empty((T, C), dtype=torch.float64, device=f"cuda:{local_rank}") for _ in range(world)]
    else:
        out = ctx.empty((T, C))

    def step():
        if args.fused:
            _, status = ctx.bcsd_fit_predict(_lib.BCSD_TAS, fields["X_hist"], fields["y_obs"], gid, 12, fields["X_fut"], gid,
                                             True, out=out)
        else:
            st = ctx.bcsd_fit(_lib.BCSD_TAS, fields["X_hist"], fields["y_obs"], gid, 12, True)
            _, status = ctx.bcsd_predict(st, fields["X_fut"], gid, out=out)
            st.close()
        if use_dist and args.gather:
            dist.gather(out_t, gather_list, dst=0)
        return status

    def barrier():
        ctx.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.prof_reset()
    ctx.prof_enable(True)  # HIP events around every kernel launch, on the engine's stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        status = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    prof = ctx.prof()
    gather_ms = None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the gather of the predicted field to rank 0, timed once outside the timed region
        try:
            barrier()
            g0 = time.perf_counter()
            dist.gather(out_t, gather_list, dst=0)
            barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
        except Exception as e:  # noqa: BLE001  (never lose the throughput line over the side measurement)
            gather_ms = f"failed: {e}"

    # ---- parity spot check: part of the cpu_baseline leg (the oracle's first run is compared with the engine's
    # output for the same cells, outside the timed region) ----
    def check_parity(exp):
        import ctypes as Cc

        n = min(args.check_cells, C, exp.shape[1])
        if n <= 0 or c_off != 0:
            return None
        rows = np.unique(np.linspace(0, T - 1, 96).astype(np.int64))  # 96 sampled rows x n cells
        got = np.empty((len(rows), n))
        for i, t in enumerate(rows):
            ctx.lib.sd_memcpy_d2h(ctx.handle, got[i].ctypes.data_as(Cc.c_void_p), Cc.c_void_p(out.ptr + int(t) * C * 8), n * 8)
        ref = exp[rows][:, :n]
        err = np.abs(got - ref)
        tol = 1e-6 * np.std(exp) + 1e-6 * np.abs(ref)
        return "ok" if bool((err <= tol).all()) and bool((status[:n] == 0).all()) else f"FAILED max_err={err.max():.3e}"

    parity = None
    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            baseline, parity = cpu_baseline(index, args.seed, c_full, args.cpu_baseline_seconds, check_parity)
        except Exception as e:  # noqa: BLE001
            parity = f"not run: {e}"

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()  # before the JSON line: RCCL may print teardown info
    if rank != 0:
        return

    ms_per_step = elapsed * 1e3 / args.steps
    value = C * world * args.steps / elapsed
    kern = {k: v["ms"] / max(1, v["launches"]) for k, v in prof.items() if k.startswith("bcsd_") and "mask" not in k}
    launches_per_step = {k: prof[k]["launches"] / args.steps for k in kern}
    kernel_ms = sum(kern[k] * launches_per_step[k] for k in kern)  # hot-path kernel time per step
    alg_bytes = float(C) * T * BYTES_PER_CELL_STEP
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # HBM bytes per step from the committed rocprofv3 PMC passes of this exact workload (separate --pmc runs,
    # gfx950 FETCH_SIZE correction calibrated on a known byte count: profiles/pmc_traffic.json); null otherwise
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        if pt["workload"]["cells"] == C and pt["workload"]["timesteps"] == T and pt["workload"]["kernel"] == "+".join(sorted(kern)):
            traffic = pt["traffic_bytes_per_step"]
    except Exception:  # noqa: BLE001
        traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "kernel": "+".join(sorted(kern)), "kernel_ms_per_step": kernel_ms,
                "algorithmic_bytes_per_step": alg_bytes,
                "per_kernel_avg_ms": kern, "launches_per_step": launches_per_step}
    line = {
        "metric": "grid-cells downscaled/sec (fit+predict), 40yr daily series",
        "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"BcsdTemperature quantile mapping, {C} cells x {T} steps per GPU (BASELINE configs[1])",
                   "cells_per_gpu": C, "timesteps": T, "groups": 12, "fused_fit_predict": bool(args.fused),
                   "gather_in_step": bool(use_dist and args.gather), "gather_to_root_ms": gather_ms, "device": info["name"]},
        "roofline": roofline,
        "parity_check": parity,
    }
    if baseline is not None:  # rank 0 at N = 1 only
        line["cpu_baseline"] = baseline
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
