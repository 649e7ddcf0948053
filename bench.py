#!/usr/bin/env python
"""Headline benchmark: grid cells downscaled per second (fit + predict), 40-year daily series.

    python bench.py                                   # BASELINE configs[1] on one MI355X
    python bench.py --config 3 | 4                    # BcsdPrecipitation 250k cells | PureAnalog k=30 100k cells
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # N = 8: BASELINE configs[4]

Workloads (BASELINE.json `configs`; float64, synthetic fields generated *in HBM* by the engine's
counter-based generator, mirror: skdownscale_amd/synth.py; inputs resident before the timed region):
  2  BcsdTemperature quantile mapping, 100 000 cells x 14 600 daily steps       (default at N = 1)
  3  BcsdPrecipitation (zero-inflated), 250 000 cells x 14 600
  4  PureAnalog(n_analogs=30, kind='mean_analogs'), F = 1, 100 000 cells x 14 600 (fit + predict)
  5  BcsdTemperature, 125 000 cells per GPU: 1 000 000 cells at N = 8            (default at N > 1; weak scaling)
One step = one full pass of the hot path over the batch: fit on (X_hist, y_obs) + predict on X_fut for every cell.

N > 1: one process per GPU; the launcher only provides RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Cells shard
across the ranks with no exchange during fit / predict; barrier, max-over-ranks timing and the gather of the predicted
field to rank 0 go through the engine's own RCCL layer (sd_comm_*: no torch import anywhere in this file).  `value`
leaves every shard resident on its GPU (how a downstream regridder or a chunked writer consumes it); a second timed loop
adds the gather to rank 0 over xGMI, chunked by cells so that a chunk's transfer overlaps the next chunk's kernels, and
reports `value_with_gather`.

Also in the line: `roofline` (algorithmic bytes / kernel time from HIP events on the engine's stream), `end_to_end` (the
host-buffer API on NumPy arrays: PCIe inclusive), `cpu_baseline` (plain-C port, OpenMP on one socket) and
`cpu_baseline_numpy` (the per-cell NumPy restatement on one core), both on bounded samples, rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "scikit-downscale_amd"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "grid-cells downscaled/sec (fit+predict), 40yr daily series"

# algorithmic bytes per (cell, time step), SURVEY.md 8(d): fields read + written once, float64
WORKLOADS = {
    2: dict(kind="bcsd_tas", cells=100_000, bytes_per_step=32,
            name="BcsdTemperature quantile mapping, {C} cells x {T} steps per GPU (BASELINE configs[1])"),
    3: dict(kind="bcsd_pr", cells=250_000, bytes_per_step=24,
            name="BcsdPrecipitation (zero-inflated), {C} cells x {T} steps per GPU (BASELINE configs[2])"),
    4: dict(kind="analog", cells=100_000, bytes_per_step=48,
            name="PureAnalog(n_analogs=30, kind='mean_analogs') F=1, {C} cells x {T} steps per GPU (BASELINE configs[3])"),
    5: dict(kind="bcsd_tas", cells=125_000, bytes_per_step=32,
            name="BcsdTemperature quantile mapping, {C} cells x {T} steps per GPU: {Ct} cells over {N} GPUs (BASELINE configs[4] at N=8)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (default: >= 5 s of GPU time at ~21 ms per step)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="BASELINE config (1-based); 0: 2 at N=1, 5 at N>1")
    ap.add_argument("--cells", type=int, default=0, help="cells per GPU (default: the config's)")
    ap.add_argument("--times", type=int, default=14_600)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gather-steps", type=int, default=3, help="N>1: steps of the second loop that includes the gather to rank 0")
    ap.add_argument("--gather-chunks", type=int, default=4, help="N>1: cell chunks whose transfers overlap the next chunk's kernels")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline, cpu_baseline_numpy and end_to_end")
    ap.add_argument("--check-cells", type=int, default=32, help="cells verified against the oracle outside the timed region")
    return ap.parse_args()


# ---- CPU baselines (rank 0, N = 1; outside every timed region) ------------------------------------------------------

def host_cpu_info():
    """model / sockets / cores from lscpu, and the hardware threads of socket 0"""
    info = {"model": None, "sockets": None, "cores_per_socket": None, "threads_per_core": None}
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            v = v.strip()
            if k == "Model name":
                info["model"] = v
            elif k == "Socket(s)":
                info["sockets"] = int(v)
            elif k == "Core(s) per socket":
                info["cores_per_socket"] = int(v)
            elif k == "Thread(s) per core":
                info["threads_per_core"] = int(v)
        cpus = []
        for line in subprocess.run(["lscpu", "-p=CPU,SOCKET"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line and not line.startswith("#"):
                cpu, sock = line.split(",")[:2]
                if sock in ("0", ""):
                    cpus.append(int(cpu))
        allowed = os.sched_getaffinity(0)
        info["socket0_cpus"] = sorted(set(cpus) & allowed) or sorted(allowed)
    except Exception:  # noqa: BLE001
        info["socket0_cpus"] = sorted(os.sched_getaffinity(0))
    return info


def cpu_baseline(index, seed, c_full, target_seconds, check=None):
    """(B) the plain-C restatement (oracle/sd_oracle.c, 'port': OpenMP over cell blocks) on the hardware threads of ONE
    socket, bounded sample; (A) the per-cell NumPy restatement (oracle/bcsd_oracle.py: the reference's steps per cell and
    month -- sort, searchsorted, interp) on one core.  The first (small) C run doubles as the checker of the engine's
    output: ``check(exp)`` receives the oracle's result for the first cells (outside the timed region)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from skdownscale_amd import synth

    cpu = host_cpu_info()
    before = os.sched_getaffinity(0)
    os.sched_setaffinity(0, cpu["socket0_cpus"])  # before the OpenMP runtime starts its threads
    try:
        import bcsd_oracle
        import c_oracle

        if not c_oracle.available():
            return None, None, None
        threads = min(c_oracle.max_threads(), len(cpu["socket0_cpus"]))
        gid = (np.asarray(index.month) - 1).astype(np.int32)

        def fields(n):
            cells = np.arange(n)
            return tuple(synth.tas_field(name, seed, index, cells, c_full) for name in ("X_hist", "y_obs", "X_fut"))

        def run(n, seconds=0.0):
            X, y, Xp = fields(n)
            spent, reps, out = 0.0, 0, None
            while reps == 0 or (spent < seconds and reps < 16):
                t0 = time.perf_counter()
                out, _ = c_oracle.bcsd_fit_predict(0, X, y, Xp, gid, gid, nthreads=threads)
                spent += time.perf_counter() - t0
                reps += 1
            return spent, reps, out

        n0 = 4 * threads
        dt, _, exp = run(n0)
        parity = check(exp) if check is not None else None
        n = int(max(n0, min(n0 / dt * target_seconds, 8192)))
        dt, reps, _ = run(n, target_seconds)
        socket_note = (f"{cpu['model']}, socket 0 of {cpu['sockets']} ({cpu['cores_per_socket']} cores x {cpu['threads_per_core']} "
                       f"threads per socket)")
        port = {"value": n * reps / dt, "unit": "cells/s", "cores": threads, "kind": "port", "cpu": socket_note,
                "sample": f"{n} cells x {len(index)} steps x {reps} passes, oracle/sd_oracle.c (OpenMP, {threads} threads pinned to "
                          f"socket 0), {dt:.1f} s"}
        # (A) one core, NumPy per cell: a few cells are enough (~0.1 s each)
        os.sched_setaffinity(0, cpu["socket0_cpus"][:1])
        X, y, Xp = fields(2048)
        t0, done = time.perf_counter(), 0
        while done < 2048 and (done == 0 or time.perf_counter() - t0 < 0.4 * target_seconds):
            bcsd_oracle.pointwise_fit_predict(0, X[:, done:done + 1], y[:, done:done + 1], Xp[:, done:done + 1], gid, gid)
            done += 1
        dtn = time.perf_counter() - t0
        numpy_leg = {"value": done / dtn, "unit": "cells/s", "cores": 1, "kind": "port", "cpu": cpu["model"],
                     "sample": f"{done} cells x {len(index)} steps, oracle/bcsd_oracle.py (per-cell NumPy loop: np.sort, "
                               f"np.searchsorted, np.interp per month), {dtn:.1f} s"}
        return port, numpy_leg, parity
    finally:
        os.sched_setaffinity(0, before)


def end_to_end(ctx, index, seed, c_full, n_cells=8192):
    """The drop-in path on host buffers (sd_bcsd_fit + sd_bcsd_predict on NumPy arrays): H2D of three fields, kernels, D2H
    of the result -- PCIe inclusive.  Bounded sample; never the headline value."""
    from skdownscale_amd import _lib, synth

    cells = np.arange(n_cells)
    X, y, Xp = (synth.tas_field(name, seed, index, cells, c_full) for name in ("X_hist", "y_obs", "X_fut"))
    gid = (np.asarray(index.month) - 1).astype(np.int32)
    def passes(n, reuse):
        times, buf = [], (np.empty_like(Xp) if reuse else None)
        for it in range(n):
            t0 = time.perf_counter()
            st = ctx.bcsd_fit(_lib.BCSD_TAS, X, y, gid, 12, True)
            out, _ = ctx.bcsd_predict(st, Xp, gid, out=buf)
            dt = time.perf_counter() - t0
            st.close()
            del out
            if it >= 2:  # the first passes create the staging buffers, the device blocks and (reuse) touch the result buffer
                times.append(dt)
        return sorted(times)[len(times) // 2], min(times)

    fresh, fresh_best = passes(5, False)   # a new NumPy result array every pass: its first-touch page faults are in the time
    reused, reused_best = passes(7, True)  # the caller's result buffer, reused (a pipeline writing one slab after the other)
    moved = 4 * X.nbytes
    return {"value": n_cells / reused, "unit": "cells/s", "cells": n_cells, "seconds": reused, "best_seconds": reused_best,
            "host_bytes_moved": moved, "effective_GBps": moved / reused / 1e9,
            "fresh_result_array": {"value": n_cells / fresh, "seconds": fresh, "best_seconds": fresh_best, "effective_GBps": moved / fresh / 1e9},
            "path": "sd_bcsd_fit + sd_bcsd_predict on pageable NumPy arrays (H2D of X_hist, y_obs, X_fut; D2H of the result), median of "
                    "the timed passes; value: result buffer reused across passes; fresh_result_array: np.empty per pass (first-touch "
                    "page faults of ~1 GB included: 4 KB pages fault at ~14 GB/s on this host unless free huge pages are at hand)"}


# ---- the benchmark ----------------------------------------------------------------------------------------------------

def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:  # started by hand: re-launch one process per GPU
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
                                   os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")

    from skdownscale_amd import _lib, synth
    from skdownscale_amd.engine import Context
    from skdownscale_amd.shard import Communicator, Rendezvous

    config = args.config or (2 if world == 1 else 5)
    wl = WORKLOADS[config]
    import ctypes

    ndev = ctypes.c_int(0)
    _lib.check(_lib.load().sd_device_count(ctypes.byref(ndev)))
    ctx = Context(local_rank % max(1, ndev.value))  # (more ranks than GPUs -- a test on a small box -- share devices)
    # control plane (barrier, max-over-ranks clock): a TCP star around rank 0; data path (gather of the predicted fields):
    # RCCL through the engine's C ABI.  The throughput line does not depend on the second one coming up.
    rdv = Rendezvous(rank, world) if world > 1 else None
    comm, comm_error = None, None
    if world > 1:
        try:
            comm = Communicator.from_env(ctx, rendezvous=rdv)
        except Exception as e:  # noqa: BLE001
            comm_error = f"{type(e).__name__}: {e}"
        if rdv.allreduce_max(0.0 if comm is not None else 1.0) > 0.0 and comm is not None:  # all ranks or none
            comm.close()
            comm, comm_error = None, "communicator creation failed on another rank"
    info = ctx.device_info()
    T = args.times
    C = args.cells or wl["cells"]
    c_full, c_off = C * world, C * rank
    index = synth.daily_calendar(T)
    gid = (np.asarray(index.month) - 1).astype(np.int32)

    def field(kind, stream, shape=(T, C), **kw):
        d = ctx.empty(shape)
        ctx.synth_fill(ctx.wrap(d.ptr, (int(np.prod(shape[:-1])), shape[-1])), kind, args.seed, stream, c_offset=c_off, c_full=c_full, **kw)
        return d

    fields = {}
    if wl["kind"] == "bcsd_tas":
        tabs = synth.tas_tables(index)
        for name in ("X_hist", "y_obs", "X_fut"):
            fields[name] = field(synth.GAUSS, tabs[name]["stream"], base=tabs[name]["base"], amp=tabs[name]["amp"],
                                 cell_scale=tabs[name]["cell_scale"])
        out = ctx.empty((T, C))

        def step(cells=None, dst=None):
            f = fields if cells is None else {k: v.cells(*cells) for k, v in fields.items()}
            _, status = ctx.bcsd_fit_predict(_lib.BCSD_TAS, f["X_hist"], f["y_obs"], gid, 12, f["X_fut"], gid, True,
                                             out=out if dst is None else dst)
            return status
    elif wl["kind"] == "bcsd_pr":
        for name in ("X_hist", "y_obs", "X_fut"):
            p = synth.PR_FIELDS[name]
            fields[name] = field(synth.PRECIP, p["stream"], amp=p["amp"], p_dry=p["p_dry"])
        out = ctx.empty((T, C))

        def step(cells=None, dst=None):
            f = fields if cells is None else {k: v.cells(*cells) for k, v in fields.items()}
            _, status = ctx.bcsd_fit_predict(_lib.BCSD_PR, f["X_hist"], f["y_obs"], gid, 12, f["X_fut"], gid, True,
                                             out=out if dst is None else dst)
            return status
    else:  # PureAnalog: X [T, 1, C], y = 2 X + noise, queries Xq
        fields["y"] = field(synth.GAUSS, 20, amp=2.0, stream2=21, amp2=1.0)
        fields["X"] = field(synth.GAUSS, 20, shape=(T, 1, C))
        fields["Xq"] = field(synth.GAUSS, 22, shape=(T, 1, C))
        out = ctx.empty((T, 3, C))

        def step(cells=None, dst=None):
            st = ctx.analog_fit(fields["X"], fields["y"])
            _, status = ctx.analog_predict(st, fields["Xq"], 30, _lib.ANALOG_MEAN, out=out)
            st.close()
            return status

    def barrier():
        ctx.synchronize()
        if rdv is not None:
            rdv.barrier()

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
    if rdv is not None:
        elapsed = rdv.allreduce_max(elapsed)

    # ---- N > 1: the same step followed by the gather of the predicted field to rank 0 (cell chunks: a chunk's transfer
    # over xGMI overlaps the next chunk's kernels) ----
    gather = None
    if comm is not None and wl["kind"] != "analog" and args.gather_steps > 0:
        try:
            from skdownscale_amd.shard import cell_partition

            nchunk = max(1, min(args.gather_chunks, C // 8))
            bounds = cell_partition(C, nchunk)
            chunk_out = [ctx.empty((T, e - s)) for s, e in bounds]
            root_bufs = [ctx.empty((T * (e - s) * world,)) if rank == 0 else None for s, e in bounds]
            cells_per_rank = [np.full(world, e - s, dtype=np.int64) for s, e in bounds]

            def gather_step():
                for i, (s, e) in enumerate(bounds):
                    step((s, e), chunk_out[i])
                    comm.gather_field(chunk_out[i], cells_per_rank[i], 0, root_bufs[i], wait=False)
                comm.wait()

            gather_step()
            barrier()
            g0 = time.perf_counter()
            for _ in range(args.gather_steps):
                gather_step()
            barrier()
            gdt = rdv.allreduce_max((time.perf_counter() - g0) / args.gather_steps)
            gather = {"value_with_gather": C * world / gdt, "ms_per_step_with_gather": gdt * 1e3, "steps": args.gather_steps,
                      "cell_chunks": nchunk, "root_layout": "[chunk][rank][T][cells of the chunk], received in place over xGMI",
                      "gathered_GB_per_step": 8.0 * T * C * (world - 1) / 1e9}
        except Exception as e:  # noqa: BLE001  (never lose the throughput line over the second measurement)
            gather = {"value_with_gather": None, "error": str(e)}
    elif world > 1 and comm is None:
        gather = {"value_with_gather": None, "error": f"no RCCL communicator: {comm_error}"}

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

    parity = baseline = numpy_leg = e2e = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            if wl["kind"] == "bcsd_tas":
                step()  # `out` holds the full-grid result again
                baseline, numpy_leg, parity = cpu_baseline(index, args.seed, c_full, args.cpu_baseline_seconds, check_parity)
            else:
                baseline, numpy_leg, _ = cpu_baseline(index, args.seed, c_full, args.cpu_baseline_seconds)
                note = " (BcsdTemperature port: the C / NumPy restatements timed here cover the BCSD path only)"
                for leg in (baseline, numpy_leg):
                    if leg:
                        leg["sample"] += note
        except Exception as e:  # noqa: BLE001
            parity = f"not run: {e}"
        try:
            for d in list(fields.values()) + [out]:
                d.free()
            ctx.release_cached()
            e2e = end_to_end(ctx, index, args.seed, c_full)
        except Exception as e:  # noqa: BLE001
            e2e = {"value": None, "error": str(e)}

    if rdv is not None:
        rdv.barrier()
        if comm is not None:
            comm.close()
        rdv.close()
    if rank != 0:
        return

    ms_per_step = elapsed * 1e3 / args.steps
    value = C * world * args.steps / elapsed
    hot = {k: v for k, v in prof.items() if "mask" not in k and "status" not in k and "nan_fill" not in k and "synth" not in k}
    kern = {k: v["ms"] / max(1, v["launches"]) for k, v in hot.items()}
    launches_per_step = {k: hot[k]["launches"] / args.steps for k in kern}
    kernel_ms = sum(kern[k] * launches_per_step[k] for k in kern)  # hot-path kernel time per step
    alg_bytes = float(C) * T * wl["bytes_per_step"]
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # HBM bytes per step from the committed rocprofv3 PMC passes of this exact workload (separate --pmc runs,
    # gfx950 FETCH_SIZE correction calibrated on a known byte count: profiles/pmc_traffic.json); null otherwise
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for pt in json.load(f)["entries"]:
                w = pt["workload"]
                if w["config"] == config and w["cells"] == C and w["timesteps"] == T and w["kernel"] == "+".join(sorted(kern)):
                    traffic = pt["traffic_bytes_per_step"]
    except Exception:  # noqa: BLE001
        traffic = None
    dominant = max(kern, key=lambda k: kern[k] * launches_per_step[k]) if kern else None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "kernel": "+".join(sorted(kern)), "dominant_kernel": dominant, "kernel_ms_per_step": kernel_ms,
                "algorithmic_bytes_per_step": alg_bytes, "algorithmic_bytes_per_cell": T * wl["bytes_per_step"],
                "per_kernel_avg_ms": kern, "launches_per_step": launches_per_step}
    line = {
        "metric": METRIC, "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": wl["name"].format(C=C, T=T, Ct=C * world, N=world), "baseline_config": config, "cells_per_gpu": C,
                   "total_cells": C * world, "timesteps": T, "groups": 12, "shards": "resident on their GPUs (value); gathered to rank 0 "
                   "(value_with_gather)" if world > 1 else "single GPU", "device": info["name"]},
        "roofline": roofline,
        "parity_check": parity,
    }
    if gather is not None:
        line.update(gather)
    if baseline is not None:  # rank 0 at N = 1 only
        line["cpu_baseline"] = baseline
    if numpy_leg is not None:
        line["cpu_baseline_numpy"] = numpy_leg
    if e2e is not None:
        line["end_to_end"] = e2e
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
