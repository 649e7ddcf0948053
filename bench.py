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
host-buffer API on NumPy arrays: PCIe inclusive), `pointwise_end_to_end` (the same through
PointWiseDownscaler on host grids), `cpu_baseline` (per config: plain-C port with OpenMP for the BCSD flavours, the NumPy restatement
on one process per core for PureAnalog; one thread / process per physical core of one socket), `cpu_baseline_numpy` (the per-cell
NumPy restatement on one core) and `cpu_baseline_numpy_socket` (the same on one process per physical core), all on bounded samples,
rank 0 at N = 1 only; `parity_check` compares the engine's output for the first cells with the oracle's for every config.
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
    # N > 1: the per-GPU workload of the N = 1 line (configs[1]: 100 000 cells), so that value(N) / value(1) is a weak-scaling
    # curve over identical per-GPU work; BASELINE configs[4] itself (1 M cells over 8 GPUs = 125 000 per GPU) is timed as a second
    # leg of the same run (`baseline_config5` in the line)
    5: dict(kind="bcsd_tas", cells=100_000, bytes_per_step=32, cells_config5=125_000,
            name="BcsdTemperature quantile mapping, {C} cells x {T} steps per GPU: {Ct} cells over {N} GPUs (the N = 1 workload per GPU; "
                 "BASELINE configs[4] -- 125 000 per GPU -- in `baseline_config5`)"),
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
    ap.add_argument("--leg-timeout", type=float, default=float(os.environ.get("SD_BENCH_LEG_TIMEOUT", "420")),
                    help="N>1: seconds the legs behind the timed loop (RCCL start-up, gathers, parity, config 5) may take together before "
                         "rank 0 prints the line with what it has and every rank exits")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline, cpu_baseline_numpy and end_to_end")
    ap.add_argument("--check-cells", type=int, default=32, help="cells verified against the oracle outside the timed region")
    ap.add_argument("--parity-only", action="store_true", help="CPU leg reduced to the parity check of the first cells (no CPU timing, no end_to_end)")
    ap.add_argument("--no-secondary", action="store_true", help="default N=1 run: skip the short config 3 / config 4 runs appended as `secondary`")
    # config 4 with another estimator / kind / feature count (the `secondary.analog_kinds` / `analog_f3` lines of the default run)
    ap.add_argument("--analog-kind", default="mean_analogs", choices=["mean_analogs", "best_analog", "weight_analogs"])
    ap.add_argument("--analog-k", type=int, default=30, help="n_analogs")
    ap.add_argument("--analog-features", type=int, default=1)
    ap.add_argument("--analog-estimator", default="pure", choices=["pure", "regression"], help="PureAnalog or AnalogRegression")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)  # internal: one process of the N-process NumPy leg
    return ap.parse_args()


# ---- CPU baselines (rank 0, N = 1; outside every timed region) ------------------------------------------------------

def host_cpu_info():
    """model / sockets / cores from lscpu; the hardware threads of socket 0 and one hardware thread per physical core of it"""
    info = {"model": None, "sockets": None, "cores_per_socket": None, "threads_per_core": None}
    allowed = os.sched_getaffinity(0)
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
        cpus, first_of_core = [], {}
        for line in subprocess.run(["lscpu", "-p=CPU,CORE,SOCKET"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line and not line.startswith("#"):
                cpu, core, sock = (line.split(",") + ["", ""])[:3]
                if sock in ("0", "") and int(cpu) in allowed:
                    cpus.append(int(cpu))
                    first_of_core.setdefault(core, int(cpu))
        info["socket0_cpus"] = sorted(cpus) or sorted(allowed)
        info["socket0_cores"] = sorted(first_of_core.values()) or sorted(allowed)
    except Exception:  # noqa: BLE001
        info["socket0_cpus"] = sorted(allowed)
        info["socket0_cores"] = sorted(allowed)
    return info


ANALOG = {"kind": "mean_analogs", "k": 30, "features": 1, "estimator": "pure"}  # config 4 as BASELINE.json words it; main() may change it
if os.environ.get("SD_BENCH_ANALOG"):  # the --cpu-worker children of a run with other analog options time the same workload
    ANALOG.update(json.loads(os.environ["SD_BENCH_ANALOG"]))


def analog_is_baseline():
    return ANALOG == {"kind": "mean_analogs", "k": 30, "features": 1, "estimator": "pure"}


def host_fields(kind, seed, T, cells, c_full):
    """the workload's synthetic fields for the given cells on the host (bit-identical mirror of the device generator)"""
    from skdownscale_amd import synth

    if kind == "analog":
        return synth.analog_fields(seed, T, cells, c_full, n_features=ANALOG["features"])  # X [T,F,C], y [T,C], Xq [T,F,C]
    if kind == "bcsd_tas":
        index = synth.daily_calendar(T)
        return tuple(synth.tas_field(name, seed, index, cells, c_full) for name in ("X_hist", "y_obs", "X_fut"))
    if kind == "bcsd_pr":
        return tuple(synth.pr_field(name, seed, T, cells, c_full) for name in ("X_hist", "y_obs", "X_fut"))
    return synth.analog_fields(seed, T, cells, c_full)  # X [T,1,C], y [T,C], Xq [T,1,C]


def numpy_cells(kind, fields, gid, lo, hi):
    """the per-cell NumPy restatement of the reference's loop (core.py:69-143) over cells [lo, hi) of `fields`"""
    if kind == "analog" and not analog_is_baseline():
        # the other kinds / AnalogRegression / F > 1: the NumPy restatement of gard.py:152-224, 273-364 (oracle/analog_oracle.py)
        import analog_oracle

        X, y, Xq = fields
        return analog_oracle.pointwise_analog(X[:, :, lo:hi], y[:, lo:hi], Xq[:, :, lo:hi], ANALOG["k"], analog_oracle.KIND_NAMES[ANALOG["kind"]],
                                              regression=ANALOG["estimator"] == "regression")
    if kind == "analog":
        X, y, Xq = fields
        try:  # the reference's own neighbour search (gard.py:58-87: sklearn KDTree, gard.py:292: tree.query(X, k))
            from sklearn.neighbors import KDTree
        except ImportError:  # brute-force restatement (~25 s per cell)
            import analog_oracle

            return analog_oracle.pointwise_analog(X[:, :, lo:hi], y[:, lo:hi], Xq[:, :, lo:hi], 30, analog_oracle.KIND_MEAN)
        out = np.empty((Xq.shape[0], 3, hi - lo))
        for c in range(lo, hi):
            _, inds = KDTree(X[:, :, c]).query(Xq[:, :, c], k=30)
            analogs = y[:, c][inds]                      # gard.py:301
            out[:, 0, c - lo] = analogs.mean(axis=1)      # gard.py:329-333 (kind='mean_analogs')
            out[:, 1, c - lo] = 1.0                       # gard.py:346
            out[:, 2, c - lo] = analogs.std(axis=1)       # gard.py:345
        return out
    import bcsd_oracle

    X, y, Xp = fields
    out, _ = bcsd_oracle.pointwise_fit_predict(0 if kind == "bcsd_tas" else 1, X[:, lo:hi], y[:, lo:hi], Xp[:, lo:hi], gid, gid)
    return out


def _numpy_worker(job):
    """one process of the N-process NumPy leg: pinned to one core, its own cells, runs until the deadline"""
    kind, seed, T, c_full, first, count, seconds, cpu = job
    try:
        os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "scikit-downscale_amd"))
    from skdownscale_amd import synth

    gid = (np.asarray(synth.daily_calendar(T).month) - 1).astype(np.int32)
    fields = host_fields(kind, seed, T, np.arange(first, first + count), c_full)
    numpy_cells(kind, fields, gid, 0, 1)  # imports, first-touch
    t0, done = time.perf_counter(), 0
    while done < count and (done == 0 or time.perf_counter() - t0 < seconds):
        numpy_cells(kind, fields, gid, done, done + 1)
        done += 1
    return done, time.perf_counter() - t0


def cpu_baseline(kind, T, seed, c_full, target_seconds, check=None, parity_only=False):
    """CPU legs of one workload, on ONE socket of the GPU box, bounded samples, outside every timed region:
      port        BCSD: the plain-C restatement (oracle/sd_oracle.c, OpenMP, contiguous cell panels per thread), one thread per
                  physical core; PureAnalog: the reference's per-cell steps (sklearn KDTree + NumPy) on one process per physical core
      numpy_1     the per-cell NumPy restatement of the reference's loop on one core
      numpy_n     the same on one process per physical core of the socket
    Returned as `port`: the faster of the socket-wide legs (NumPy's vectorised np.sort beats the scalar C merge sort at scale).
    The oracle's result for the first cells is handed to ``check`` (parity of the engine's output for the same cells)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from skdownscale_amd import synth

    cpu = host_cpu_info()
    cores = cpu["socket0_cores"]
    before = os.sched_getaffinity(0)
    gid = (np.asarray(synth.daily_calendar(T).month) - 1).astype(np.int32)
    socket_note = (f"{cpu['model']}, socket 0 of {cpu['sockets']} ({cpu['cores_per_socket']} cores x {cpu['threads_per_core']} "
                   f"threads per socket), one thread / process per physical core: {len(cores)}")
    port = numpy_1 = numpy_n = parity = c_port = None
    try:
        os.sched_setaffinity(0, cores)  # before the OpenMP runtime starts its threads
        # ---- parity of the engine's output for the first cells (small oracle run) ----
        n_chk = (16 if analog_is_baseline() else 2) if kind == "analog" else 64
        chk_fields = host_fields(kind, seed, T, np.arange(n_chk), c_full)
        if kind == "analog":
            exp = numpy_cells(kind, chk_fields, gid, 0, n_chk)
        else:
            import c_oracle

            exp = (c_oracle.bcsd_fit_predict(0 if kind == "bcsd_tas" else 1, *chk_fields, gid, gid, nthreads=len(cores))[0]
                   if c_oracle.available() else numpy_cells(kind, chk_fields, gid, 0, n_chk))
        parity = check(exp) if check is not None else None
        if parity_only:
            return None, None, None, parity, None
        # ---- port ----
        if kind != "analog":
            import c_oracle

            if c_oracle.available():
                threads = min(c_oracle.max_threads(), len(cores))
                k = 0 if kind == "bcsd_tas" else 1

                base = host_fields(kind, seed, T, np.arange(1024), c_full)  # (generating the fields on the host is slow:

                def run(n, seconds):                                          #  1 024 distinct cells, repeated up to n)
                    f = tuple(np.ascontiguousarray(np.tile(a, (1, (n + 1023) // 1024))[:, :n]) for a in base)
                    spent, reps = 0.0, 0
                    while reps == 0 or (spent < seconds and reps < 32):
                        t0 = time.perf_counter()
                        c_oracle.bcsd_fit_predict(k, *f, gid, gid, nthreads=threads)
                        spent += time.perf_counter() - t0
                        reps += 1
                    return spent, reps

                n0 = 64 * threads  # one 64-cell panel per thread
                dt, _ = run(n0, 0.0)
                n = int(max(n0, min(n0 / dt * target_seconds, 16384))) // (64 * threads) * (64 * threads)
                dt, reps = run(n, target_seconds)
                port = {"value": n * reps / dt, "unit": "cells/s", "cores": threads, "kind": "port", "cpu": socket_note,
                        "sample": f"{n} cells x {T} steps x {reps} passes, oracle/sd_oracle.c ({'BcsdTemperature' if k == 0 else 'BcsdPrecipitation'} "
                                  f"flavour; OpenMP, {threads} threads = one per physical core of socket 0, static 64-cell panels), {dt:.1f} s"}
        # ---- NumPy on one core ----
        os.sched_setaffinity(0, cores[:1])
        n1 = 256
        f1 = host_fields(kind, seed, T, np.arange(n1), c_full)
        numpy_cells(kind, f1, gid, 0, 1)
        t0, done = time.perf_counter(), 0
        while done < n1 and (done == 0 or time.perf_counter() - t0 < 0.3 * target_seconds):
            numpy_cells(kind, f1, gid, done, done + 1)
            done += 1
        dt1 = time.perf_counter() - t0
        what = ("per-cell sklearn KDTree.query(k=30) + NumPy mean / std of the analogs (what gard.py:58-87, 273-346 does per cell)"
                if kind == "analog" and analog_is_baseline()
                else f"oracle/analog_oracle.py per cell ({ANALOG['estimator']}, kind={ANALOG['kind']}, k={ANALOG['k']}, F={ANALOG['features']})" if kind == "analog"
                else "oracle/bcsd_oracle.py (per-cell NumPy loop: np.sort, np.searchsorted, np.interp per month)")
        numpy_1 = {"value": done / dt1, "unit": "cells/s", "cores": 1, "kind": "port", "cpu": cpu["model"],
                   "sample": f"{done} cells x {T} steps, {what}, {dt1:.1f} s"}
        # ---- NumPy on one process per physical core ----
        os.sched_setaffinity(0, cores)
        per = max(4, int(done / dt1 * 0.5 * target_seconds) + 1)
        jobs = [(kind, seed, T, c_full, i * per, per, 0.5 * target_seconds, c) for i, c in enumerate(cores)]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", ",".join(str(v) for v in job)],
                                  stdout=subprocess.PIPE, text=True) for job in jobs]  # plain child processes: no GPU state inherited
        res = []
        for pr in procs:
            o, _ = pr.communicate(timeout=120 + 4 * target_seconds)
            d, e = o.strip().split()[-2:]
            res.append((int(d), float(e)))
        tot = sum(r[0] for r in res)
        wall = max(r[1] for r in res)
        numpy_n = {"value": tot / wall, "unit": "cells/s", "cores": len(cores), "kind": "port", "cpu": socket_note,
                   "sample": f"{tot} cells x {T} steps over {len(cores)} processes (each pinned to one physical core of socket 0, "
                             f"{wall:.1f} s), {what}"}
        c_port = port
        if port is None or numpy_n["value"] > port["value"]:  # the faster of the two socket-wide legs is the baseline
            port = dict(numpy_n)
    finally:
        os.sched_setaffinity(0, before)
    return port, numpy_1, numpy_n, parity, c_port


def oracle_parity(X, y, Xp, gid, results, statuses=None):
    """the host-path results (each [T, n] for the first n cells) against the oracle's fit + predict of the same cells (C port when
    built, else the NumPy restatement): 'ok' or a failure note; outside every timed region"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle

        if c_oracle.available():
            exp, _ = c_oracle.bcsd_fit_predict(0, X, y, Xp, gid, gid, nthreads=min(8, c_oracle.max_threads()))
        else:
            import bcsd_oracle

            exp, _ = bcsd_oracle.pointwise_fit_predict(0, X, y, Xp, gid, gid)
        tol = 1e-6 * np.nanstd(exp) + 1e-6 * np.abs(exp)
        worst = 0.0
        for i, got in enumerate(results):
            got = np.asarray(got, dtype=np.float64)
            if got.shape != exp.shape:
                return f"FAILED shape {got.shape} != {exp.shape}"
            err = np.abs(got - exp)
            if not bool((err <= tol).all()):
                return f"FAILED max_err={np.nanmax(err):.3e} (result {i})"
            if statuses is not None and not bool((np.asarray(statuses[i]) == 0).all()):
                return f"FAILED cell status (result {i})"
            worst = max(worst, float(np.nanmax(err)))
        return f"ok ({exp.shape[1]} cells x {exp.shape[0]} steps, max abs err {worst:.2e})"
    except Exception as e:  # noqa: BLE001
        return f"not run: {type(e).__name__}: {e}"


def end_to_end(ctx, index, seed, c_full, n_cells=8192):
    """The drop-in path on host buffers (sd_bcsd_fit + sd_bcsd_predict on NumPy arrays): H2D of three fields, kernels, D2H
    of the result -- PCIe inclusive.  Bounded sample; never the headline value."""
    from skdownscale_amd import _lib, synth

    cells = np.arange(n_cells)
    X, y, Xp = (synth.tas_field(name, seed, index, cells, c_full) for name in ("X_hist", "y_obs", "X_fut"))
    gid = (np.asarray(index.month) - 1).astype(np.int32)
    n_chk = min(16, n_cells)
    checked = {}

    def passes(n, reuse):
        times, buf = [], (np.empty_like(Xp) if reuse else None)
        for it in range(n):
            t0 = time.perf_counter()
            st = ctx.bcsd_fit(_lib.BCSD_TAS, X, y, gid, 12, True)
            out, status = ctx.bcsd_predict(st, Xp, gid, out=buf)
            dt = time.perf_counter() - t0
            st.close()
            if it == n - 1:  # (outside the timed region) the last pass's result for the first cells, for the parity check below
                checked["reuse" if reuse else "fresh"] = (np.array(out[:, :n_chk]), np.array(status[:n_chk]))
            del out
            if it >= 2:  # the first passes create the staging buffers, the device blocks and (reuse) touch the result buffer
                times.append(dt)
        return sorted(times)[len(times) // 2], min(times)

    def passes32(n):  # the same grid as float32: 4 bytes per sample over PCIe, widened / narrowed on the device
        X32, y32, Xp32 = X.astype(np.float32), y.astype(np.float32), Xp.astype(np.float32)
        times = []
        for it in range(n):
            t0 = time.perf_counter()
            st = ctx.bcsd_fit(_lib.BCSD_TAS, X32, y32, gid, 12, True)
            out, _ = ctx.bcsd_predict(st, Xp32, gid, out_dtype=np.float32)
            dt = time.perf_counter() - t0
            st.close()
            del out
            if it >= 2:
                times.append(dt)
        return sorted(times)[len(times) // 2]

    fresh, fresh_best = passes(5, False)   # a new NumPy result array every pass: its first-touch page faults are in the time
    reused, reused_best = passes(7, True)  # the caller's result buffer, reused (a pipeline writing one slab after the other)
    moved = 4 * X.nbytes
    parity = oracle_parity(X[:, :n_chk], y[:, :n_chk], Xp[:, :n_chk], gid, [checked[k][0] for k in ("fresh", "reuse")],
                           [checked[k][1] for k in ("fresh", "reuse")])
    try:
        f32 = passes32(5)
        f32 = {"value": n_cells / f32, "seconds": f32, "note": "float32 host grids: float32 over PCIe (half the bytes), float64 arithmetic on the device"}
    except Exception as e:  # noqa: BLE001
        f32 = {"value": None, "error": str(e)}
    return {"value": n_cells / reused, "unit": "cells/s", "cells": n_cells, "seconds": reused, "best_seconds": reused_best, "float32_grids": f32,
            "parity_check": parity, "host_bytes_moved": moved, "effective_GBps": moved / reused / 1e9,
            "fresh_result_array": {"value": n_cells / fresh, "seconds": fresh, "best_seconds": fresh_best, "effective_GBps": moved / fresh / 1e9},
            "path": "sd_bcsd_fit + sd_bcsd_predict on pageable NumPy arrays (H2D of X_hist, y_obs, X_fut; D2H of the result), median of "
                    "the timed passes; value: result buffer reused across passes; fresh_result_array: np.empty per pass (first-touch "
                    "page faults of ~1 GB included: 4 KB pages fault at ~14 GB/s on this host unless free huge pages are at hand)"}


def pointwise_end_to_end(index, seed, c_full, n_cells=8192):
    """The drop-in surface a user of the reference calls: PointWiseDownscaler(BcsdTemperature()).fit(X, y).predict(X_fut) on
    host arrays of one grid (core.py:198-336) -- wrappers, validation, H2D / D2H copies and a fitted state included.
    Bounded sample; never the headline value."""
    import pandas as pd  # noqa: F401  (the estimators take a DatetimeIndex)

    from skdownscale_amd import BcsdTemperature, PointWiseDownscaler, synth
    from skdownscale_amd.core import GridArray

    cells = np.arange(n_cells)
    ny = 64
    nx = n_cells // ny
    X, y, Xp = (synth.tas_field(name, seed, index, cells, c_full).reshape(len(index), ny, nx) for name in ("X_hist", "y_obs", "X_fut"))
    dims = ("time", "y", "x")
    mk = lambda a: GridArray(a, dims, {"time": index})  # noqa: E731
    Xg, yg, Xpg = mk(X), mk(y), mk(Xp)
    times = []
    for it in range(5):
        t0 = time.perf_counter()
        model = PointWiseDownscaler(BcsdTemperature(return_anoms=True))
        model.fit(Xg, yg)
        res = model.predict(Xpg)
        field = np.asarray(res.values if hasattr(res, "values") else res)
        dt = time.perf_counter() - t0
        if it == 4:  # (outside the timed region) the first cells of the last pass, for the parity check below
            first = np.array(field.reshape(len(index), -1)[:, :16])
        # (released outside the timed region, like `out` in end_to_end: until round 4 the rebinding of a throw-away name freed the
        # previous pass's 1 GB result *inside* it -- 42 ms of munmap per pass, profiles/r04/prof_pointwise2.log)
        del model, res, field
        if it >= 2:
            times.append(dt)
    med = sorted(times)[len(times) // 2]
    moved = 4 * X.nbytes
    T = len(index)
    gid = (np.asarray(index.month) - 1).astype(np.int32)
    parity = oracle_parity(*(np.ascontiguousarray(a.reshape(T, -1)[:, :16]) for a in (X, y, Xp)), gid, [first])
    return {"value": n_cells / med, "unit": "cells/s", "cells": n_cells, "seconds": med, "best_seconds": min(times),
            "parity_check": parity, "effective_GBps": moved / med / 1e9,
            "path": "PointWiseDownscaler(BcsdTemperature()).fit(X, y) + .predict(X_fut) on host GridArrays [time, 64, 128] "
                    "(estimator wrappers, validation, H2D of three fields, fitted state, D2H of the result), median of 3 timed passes"}


# ---- the benchmark ----------------------------------------------------------------------------------------------------

def main():
    args = parse()
    if args.cpu_worker:
        kind, seed, T, c_full, first, count, seconds, cpu = args.cpu_worker.split(",")
        done, dt = _numpy_worker((kind, int(seed), int(T), int(c_full), int(first), int(count), float(seconds), int(cpu)))
        print(done, dt)
        return
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
    wl = dict(WORKLOADS[config])
    if wl["kind"] == "analog":
        ANALOG.update(kind=args.analog_kind, k=args.analog_k, features=args.analog_features, estimator=args.analog_estimator)
        os.environ["SD_BENCH_ANALOG"] = json.dumps(ANALOG)  # inherited by the --cpu-worker processes of cpu_baseline
        if not analog_is_baseline():
            F = args.analog_features
            wl["bytes_per_step"] = 8 * (F + 1 + F + 3)  # X, y, Xq read, the three output columns written
            est = (f"AnalogRegression(n_analogs={args.analog_k})" if args.analog_estimator == "regression"
                   else f"PureAnalog(n_analogs={args.analog_k}, kind='{args.analog_kind}')")
            wl["name"] = est + f" F={F}, {{C}} cells x {{T}} steps per GPU"
    import ctypes

    ndev = ctypes.c_int(0)
    _lib.check(_lib.load().sd_device_count(ctypes.byref(ndev)))
    ctx = Context(local_rank % max(1, ndev.value))  # (more ranks than GPUs -- a test on a small box -- share devices)
    # control plane (barrier, max-over-ranks clock): a TCP star around rank 0; data path (gather of the predicted fields):
    # RCCL through the engine's C ABI.  The throughput line does not depend on the second one coming up.
    # (RCCL comes up BEHIND the timed loop -- see the watchdog there: the path has never run on N > 1 GPUs of one node before the
    # driver's scaling run, and a communicator that hangs while it is created must not cost the throughput line)
    rdv = Rendezvous(rank, world) if world > 1 else None
    comm, comm_error = None, None
    info = ctx.device_info()
    T = args.times
    C = args.cells or wl["cells"]
    c_full, c_off = C * world, C * rank
    index = synth.daily_calendar(T)
    gid = (np.asarray(index.month) - 1).astype(np.int32)

    def field(kind, stream, shape=(T, C), offs=None, **kw):
        d = ctx.empty(shape)
        o, f = (c_off, c_full) if offs is None else offs
        ctx.synth_fill(ctx.wrap(d.ptr, (int(np.prod(shape[:-1])), shape[-1])), kind, args.seed, stream, c_offset=o, c_full=f, **kw)
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
        from skdownscale_amd.engine import DeviceArray

        F = ANALOG["features"]
        fields["y"] = field(synth.GAUSS, 20, amp=2.0, stream2=21, amp2=1.0)
        for name, s0 in (("X", 20), ("Xq", 22)):  # [T, F, C]: feature f of time t is row t * F + f (streams as synth.analog_fields)
            fields[name] = ctx.empty((T, F, C))
            for f in range(F):
                view = DeviceArray(ctx, (T, C), dptr=fields[name].ptr + f * C * 8, owner=False, ld=F * C, base=fields[name])
                ctx.synth_fill(view, synth.GAUSS, args.seed, s0 + 100 * f, c_offset=c_off, c_full=c_full)
        out = ctx.empty((T, 3, C))
        k_eff = 1 if ANALOG["kind"] == "best_analog" else ANALOG["k"]  # gard.py:291-296: best_analog queries one neighbour
        kind_code = {"best_analog": _lib.ANALOG_BEST, "weight_analogs": _lib.ANALOG_WEIGHT, "mean_analogs": _lib.ANALOG_MEAN}[ANALOG["kind"]]

        def step(cells=None, dst=None):
            if ANALOG["estimator"] == "regression":  # AnalogBase.fit + AnalogRegression.predict (gard.py:58-87, 152-224)
                st = ctx.analog_fit(fields["X"], fields["y"])
                _, status = ctx.analogreg_predict(st, fields["Xq"], ANALOG["k"], out=out)
                st.close()
                return status
            if not analog_is_baseline():
                _, status = ctx.analog_fit_predict(fields["X"], fields["y"], fields["Xq"], k_eff, kind_code, out=out)
                return status
            if os.environ.get("SD_BENCH_ANALOG_SPLIT"):  # the two calls (a fitted state written, read back and dropped)
                st = ctx.analog_fit(fields["X"], fields["y"])
                _, status = ctx.analog_predict(st, fields["Xq"], 30, _lib.ANALOG_MEAN, out=out)
                st.close()
                return status
            # fit + predict in one call: the step keeps no fitted state (sd_analog_fit_predict_dev)
            _, status = ctx.analog_fit_predict(fields["X"], fields["y"], fields["Xq"], 30, _lib.ANALOG_MEAN, out=out)
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

    # ---- everything below is outside the timed region.  N > 1: the legs that follow bring up RCCL and move fields between GPUs for
    # the first time on this node; a watchdog bounds them -- when they stall, rank 0 prints the line with what it has (the stalled
    # leg named in `legs_timed_out`) and every rank leaves, so a hang in a collective cannot lose the measurement above ----
    gather = None
    parity = baseline = numpy_1 = numpy_n = e2e = pw_e2e = c_port = None
    rccl = None
    config5 = None
    def build_line():
        """the bench line from what has been measured so far (called once at the end -- or by the N > 1 watchdog while a leg stalls)"""
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
        traffic, traffic_source = None, None
        try:
            import hashlib

            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                doc = json.load(f)
            csrc = os.path.join(ROOT, "scikit-downscale_amd", "csrc")
            now = hashlib.sha256(b"".join(open(os.path.join(csrc, fn), "rb").read() for fn in sorted(os.listdir(csrc))
                                          if fn.endswith((".hip", ".h")))).hexdigest()[:16]
            src = doc.get("source", {})
            traffic_source = {"file": "profiles/pmc_traffic.json", "head": src.get("head"), "generated": src.get("generated"),
                              "kernel_sources_match": src.get("kernel_sources_sha16") == now, "entry": None}
            for pt in doc["entries"]:
                w = pt["workload"]
                if w["config"] == config and w["cells"] == C and w["timesteps"] == T:
                    if w["kernel"] == "+".join(sorted(kern)):
                        traffic = pt["traffic_bytes_per_step"]
                        traffic_source["entry"] = "matches this run's workload and kernel set"
                    else:
                        traffic_source["entry"] = f"stale: measured for kernels {w['kernel']}, this run launches {'+'.join(sorted(kern))}"
            if traffic_source["entry"] is None:
                traffic_source["entry"] = "no PMC pass for this config / size"
            if traffic is not None and not traffic_source["kernel_sources_match"]:
                traffic_source["entry"] += "; kernel sources changed since the PMC pass (refresh with tools/dev/refresh_profiles.sh)"
        except Exception as e:  # noqa: BLE001
            traffic, traffic_source = None, {"file": "profiles/pmc_traffic.json", "entry": f"unreadable: {e}"}
        dominant = max(kern, key=lambda k: kern[k] * launches_per_step[k]) if kern else None
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                    "traffic": traffic, "traffic_source": traffic_source, "kernel": "+".join(sorted(kern)), "dominant_kernel": dominant, "kernel_ms_per_step": kernel_ms,
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
        if world > 1:
            line["rccl"] = rccl
            line["baseline_config5"] = config5
            line["scaling_claim"] = ("north_star's '>= 7x at 8 GPUs vs 1' is claimed on `value`: fit + predict of every rank's shard with the "
                                     "predicted shards left resident on their GPUs (how a regridder or chunked writer consumes them); "
                                     "`value_with_gather` (all shards into rank 0's HBM over xGMI, bound by the root's 7 links) and "
                                     "`value_with_host_gather` (every rank to its own host memory over its own PCIe link) are reported beside "
                                     "it and are NOT expected to reach 7x (DESIGN.md section 5)")
        if baseline is not None:  # rank 0 at N = 1 only
            line["cpu_baseline"] = baseline
        if numpy_1 is not None:
            line["cpu_baseline_numpy"] = numpy_1
        if numpy_n is not None:
            line["cpu_baseline_numpy_socket"] = numpy_n
        if c_port is not None:
            line["cpu_baseline_c_port"] = c_port
        if e2e is not None:
            line["end_to_end"] = e2e
        if pw_e2e is not None:
            line["pointwise_end_to_end"] = pw_e2e
        return line

    leg = {"name": "start", "done": False}
    watchdog = None
    if world > 1 and args.leg_timeout > 0:
        import threading

        def on_timeout():
            if leg["done"]:
                return
            if rank == 0:
                try:
                    line = build_line()
                    line["legs_timed_out"] = {"leg": leg["name"], "after_seconds": args.leg_timeout,
                                              "note": "the legs behind the timed loop stalled; `value` was measured before them"}
                    if line.get("rccl") is None:
                        line["rccl"] = {"ranks": None, "error": f"stalled in leg '{leg['name']}'", "world_size_env": world}
                    print(json.dumps(line), flush=True)
                finally:
                    os._exit(0)
            os._exit(0)

        watchdog = threading.Timer(args.leg_timeout, on_timeout)
        watchdog.daemon = True
        watchdog.start()
    if os.environ.get("SD_BENCH_FAKE_STALL") and world > 1:  # (test hook: a leg that never returns)
        leg["name"] = "fake stall (SD_BENCH_FAKE_STALL)"
        time.sleep(1e6)
    if world > 1:
        leg["name"] = "RCCL communicator (ncclCommInitRank)"
        try:
            comm = Communicator.from_env(ctx, rendezvous=rdv)
        except Exception as e:  # noqa: BLE001
            comm_error = f"{type(e).__name__}: {e}"
        if rdv.allreduce_max(0.0 if comm is not None else 1.0) > 0.0 and comm is not None:  # all ranks or none
            comm.close()
            comm, comm_error = None, "communicator creation failed on another rank"
        if comm is not None:  # what RCCL itself says it is running on (ncclCommCount: the ranks IT sees, not WORLD_SIZE)
            try:
                rccl = dict(comm.rccl_info(), world_size_env=world)
            except Exception as e:  # noqa: BLE001
                rccl = {"ranks": None, "error": f"{type(e).__name__}: {e}", "world_size_env": world}
        else:
            rccl = {"ranks": 0, "error": comm_error, "world_size_env": world}

    # ---- N > 1: the same step followed by the gather of the predicted field to rank 0 (cell chunks: a chunk's transfer
    # over xGMI overlaps the next chunk's kernels) ----
    leg["name"] = "gather to rank 0 over xGMI (RCCL send / recv)"
    if comm is not None and wl["kind"] != "analog" and args.gather_steps > 0:
        chunk_out, root_bufs = [], []
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
        finally:  # rank 0 holds world x the shard here (117 GB at N = 8): give it back before the next leg allocates
            try:
                ctx.synchronize()
                for b in root_bufs + chunk_out:
                    if b is not None:
                        b.free()
            except Exception:  # noqa: BLE001
                pass
    elif world > 1 and comm is None:
        gather = {"value_with_gather": None, "error": f"no RCCL communicator: {comm_error}"}

    # ---- N > 1: the other sink -- every rank copies its own shard to host memory over its own PCIe link (N links in
    # parallel instead of N - 1 xGMI links into one GPU); cell chunks through two alternating host buffers ----
    leg["name"] = "shards to host memory"
    if world > 1 and wl["kind"] != "analog" and args.gather_steps > 0:
        host = None
        try:
            import ctypes as Cc

            from skdownscale_amd.shard import cell_partition

            nchunk = max(1, min(args.gather_chunks, C // 8))
            bounds = cell_partition(C, nchunk)
            dev = [ctx.empty((T, e - s)) for s, e in bounds]
            widest = max(e - s for s, e in bounds)
            hostbuf = [np.zeros(T * widest) for _ in range(2)]  # two landing buffers, alternating (pages touched outside the timed region)

            def drain(i):
                n = T * (bounds[i][1] - bounds[i][0]) * 8
                ctx.lib.sd_memcpy_d2h(ctx.handle, hostbuf[i % 2].ctypes.data_as(Cc.c_void_p), dev[i].vptr, n)

            def host_step():
                for i, (s, e) in enumerate(bounds):
                    step((s, e), dev[i])  # (returns when the chunk's kernels are done: the status comes back with it)
                    drain(i)              # calls on one context are serialised: no copy thread (the copy is 10 x the compute anyway)

            host_step()
            barrier()
            h0 = time.perf_counter()
            for _ in range(args.gather_steps):
                host_step()
            barrier()
            hdt = rdv.allreduce_max((time.perf_counter() - h0) / args.gather_steps)
            host = {"value_with_host_gather": C * world / hdt, "ms_per_step_with_host_gather": hdt * 1e3,
                    "host_gather": f"every rank's [T, {C}] shard to its process's host memory over its own PCIe link, {nchunk} cell chunks, "
                                   "compute and copy of a chunk back to back (one context: serialised calls)",
                    "host_GB_per_step_per_rank": 8.0 * T * C / 1e9}
        except Exception as e:  # noqa: BLE001
            host = {"value_with_host_gather": None, "host_gather_error": f"{type(e).__name__}: {e}"}
        gather = dict(gather or {}, **host)

    # ---- parity spot check: part of the cpu_baseline leg (the oracle's result for the first cells is compared with the
    # engine's output for the same cells, outside the timed region) ----
    def check_parity(exp):
        import ctypes as Cc

        n = min(args.check_cells, C, exp.shape[-1])
        if n <= 0 or c_off != 0:
            return None
        rows_per_t = 3 if wl["kind"] == "analog" else 1                # `out` is [T, C] or [T, 3, C]
        ts = np.unique(np.linspace(0, T - 1, 96).astype(np.int64))      # 96 sampled time steps x n cells
        exp2 = exp.reshape(T * rows_per_t, exp.shape[-1])
        rows = (ts[:, None] * rows_per_t + np.arange(rows_per_t)[None, :]).ravel()
        got = np.empty((len(rows), n))
        for i, r in enumerate(rows):
            ctx.lib.sd_memcpy_d2h(ctx.handle, got[i].ctypes.data_as(Cc.c_void_p), Cc.c_void_p(out.ptr + int(r) * C * 8), n * 8)
        ref = exp2[rows][:, :n]
        err = np.abs(got - ref)
        tol = 1e-6 * np.nanstd(exp) + 1e-6 * np.abs(ref)
        ok = bool((err <= tol).all()) and bool((np.asarray(status)[:n] == 0).all())
        return "ok" if ok else f"FAILED max_err={np.nanmax(err):.3e}"

    leg["name"] = "parity check of rank 0's shard"
    if rank == 0 and world > 1 and not args.no_cpu_baseline:
        # N > 1: rank 0's shard (cells 0 .. C-1 of the C * world grid) against the oracle, outside every timed region; the
        # cpu_baseline legs themselves stay with the N = 1 line
        try:
            step()
            ctx.synchronize()
            _, _, _, parity, _ = cpu_baseline(wl["kind"], T, args.seed, c_full, 0.0, check_parity, parity_only=True)
        except Exception as e:  # noqa: BLE001
            parity = f"not run: {type(e).__name__}: {e}"
    leg["name"] = "baseline_config5 (125 000 cells per GPU)"
    if world > 1 and wl.get("cells_config5") and not args.cells and wl["kind"] == "bcsd_tas":
        # BASELINE configs[4] at its own size: 125 000 cells per GPU (1 M cells at N = 8), shards resident
        try:
            if rdv is not None:
                rdv.barrier()  # (rank 0 has finished its parity check)
            C5 = int(wl["cells_config5"])
            for d in list(fields.values()) + [out]:
                d.free()
            ctx.release_cached()
            tabs = synth.tas_tables(index)
            f5 = {name: field(synth.GAUSS, tabs[name]["stream"], shape=(T, C5), offs=(C5 * rank, C5 * world), base=tabs[name]["base"],
                              amp=tabs[name]["amp"], cell_scale=tabs[name]["cell_scale"]) for name in ("X_hist", "y_obs", "X_fut")}
            out5 = ctx.empty((T, C5))
            fields, out = f5, out5  # (released with the rest below)

            def step5():
                ctx.bcsd_fit_predict(_lib.BCSD_TAS, f5["X_hist"], f5["y_obs"], gid, 12, f5["X_fut"], gid, True, out=out5)

            n5 = max(1, min(args.steps, 20))
            for _ in range(2):
                step5()
            barrier()
            t5 = time.perf_counter()
            for _ in range(n5):
                step5()
            barrier()
            dt5 = rdv.allreduce_max(time.perf_counter() - t5)
            config5 = {"value": C5 * world * n5 / dt5, "unit": "cells/s", "cells_per_gpu": C5, "total_cells": C5 * world, "steps": n5,
                       "ms_per_step": dt5 * 1e3 / n5, "workload": "BASELINE configs[4]: BcsdTemperature, 125 000 cells per GPU "
                       f"({C5 * world} cells over {world} GPUs; 1 M at N = 8), shards resident"}
        except Exception as e:  # noqa: BLE001
            config5 = {"value": None, "error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            step()  # `out` holds the full-grid result again
            ctx.synchronize()
            baseline, numpy_1, numpy_n, parity, c_port = cpu_baseline(wl["kind"], T, args.seed, c_full, args.cpu_baseline_seconds, check_parity,
                                                                       parity_only=args.parity_only)
        except Exception as e:  # noqa: BLE001
            parity = f"not run: {type(e).__name__}: {e}"
        try:
            for d in list(fields.values()) + [out]:
                d.free()
            ctx.release_cached()
            if wl["kind"] == "bcsd_tas" and not args.parity_only:
                e2e = end_to_end(ctx, index, args.seed, c_full)
                pw_e2e = pointwise_end_to_end(index, args.seed, c_full)
        except Exception as e:  # noqa: BLE001
            e2e = {"value": None, "error": str(e)}

    leg["name"] = "closing barrier"
    if rdv is not None:
        rdv.barrier()
        if comm is not None:
            comm.close()
        rdv.close()
    leg["done"] = True
    if watchdog is not None:
        watchdog.cancel()
    if rank != 0:
        return
    line = build_line()
    # ---- the other single-GPU configurations of BASELINE.json, a few steps each, so that the driver's one run times them too:
    # each in a process of its own (this one has released its fields), full size, parity of the first cells against the oracle
    if world == 1 and config == 2 and not args.no_cpu_baseline and not args.parity_only and not args.no_secondary and not args.cells \
            and args.times == 14_600:
        secondary = {}
        for c2, steps in ((3, 10), (4, 8)):
            try:
                o = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", str(c2), "--steps", str(steps), "--warmup", "2",
                                    "--parity-only", "--seed", str(args.seed)], capture_output=True, text=True, timeout=600)
                d = json.loads(o.stdout.strip().splitlines()[-1])
                secondary[f"config{c2}"] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "steps": d["steps"],
                                            "ms_per_step": d["ms_per_step"], "parity_check": d["parity_check"],
                                            "roofline": {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel",
                                                                                       "kernel_ms_per_step", "algorithmic_bytes_per_step")}}
            except Exception as e:  # noqa: BLE001
                secondary[f"config{c2}"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        # the rest of the analog surface at BASELINE size (F = 1): the reference's default PureAnalog() (kind='best_analog',
        # n_analogs=200: one neighbour, gard.py:257-271, 291-296), weight_analogs, AnalogRegression; and F = 3 (KDTree-equivalent
        # neighbour search in three dimensions; cells sized to a few seconds)
        def analog_line(extra, steps, warmup):
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "4", "--steps", str(steps), "--warmup", str(warmup),
                                "--parity-only", "--seed", str(args.seed)] + extra, capture_output=True, text=True, timeout=900)
            d = json.loads(o.stdout.strip().splitlines()[-1])
            r = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "ms_per_step": d["ms_per_step"],
                 "parity_check": d["parity_check"],
                 "roofline": {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms_per_step",
                                                            "algorithmic_bytes_per_step")}}
            return r, d

        kinds = {}
        for key, extra, steps in (("best_analog_n200", ["--analog-kind", "best_analog", "--analog-k", "200"], 6),
                                  ("weight_analogs_k30", ["--analog-kind", "weight_analogs"], 4),
                                  ("analog_regression_k30", ["--analog-estimator", "regression"], 4)):
            try:
                kinds[key], _ = analog_line(extra, steps, 1)
            except Exception as e:  # noqa: BLE001
                kinds[key] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        secondary["analog_kinds"] = kinds
        try:
            f3, d = analog_line(["--analog-features", "3", "--cells", "16384"], 2, 1)
            cells_s = d["value"]
            # the brute-force equivalent: 3 F Tf Tq flops per cell at the measured cells per second
            f3["brute_force_equivalent_tflops"] = cells_s * 3.0 * 3 * args.times * args.times / 1e12
            # against the float64 vector peak of the guide (78.6 TFLOP/s): what a full pairwise scan in float64 would need at this rate
            f3["brute_force_equivalent_frac_of_fp64_vector_peak"] = f3["brute_force_equivalent_tflops"] / 78.6
            f3["note"] = ("KDTree-equivalent exact search: slab scan over feature 0; per chunk of 64 points a float32 pre-filter on the matrix cores "
                          "(v_mfma_f32_32x32x2_f32), exact float64 distances for the flagged points only, candidate lists pruned by a register sorting "
                          "network (analog_slab_topk_kernel); flops of a full float64 pairwise scan at this rate are reported for scale, the kernel does "
                          "not execute them; roofline.frac is the HBM fraction of the algorithmic bytes")
            secondary["analog_f3"] = f3
        except Exception as e:  # noqa: BLE001
            secondary["analog_f3"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        line["secondary"] = secondary
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
