#!/usr/bin/env python
"""Randomised sweep of bcsd_fd_kernel (round 6: whole-lane months, tiles by LDS-DMA) -- development library:
random calendars whose groups are whole lanes of 20 samples (1 160 .. 1 240, mixed with groups that are not), even cell counts with
ragged last tiles, cell views of wider fields, continuous / dyadic (exact ties) / nearly tied / constant-stretch data, masked and
non-finite cells; the result must equal the register-tile kernel (SD_FX_NODMA) bit for bit, both schedules of the DMA kernel, and the
NumPy oracle within 1e-6 relative for a few cells.   usage: fuzz_fd.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scikit-downscale_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import bcsd_oracle as bo  # noqa: E402
from _cases import assert_close  # noqa: E402
from skdownscale_amd import _lib  # noqa: E402
from skdownscale_amd.engine import Context  # noqa: E402


def main(n_cases, seed):
    ctx = Context(0, lib_path=_lib.DEV_LIB_PATH)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    stats = {"cases": 0, "dma_launches": 0, "worklist_styles": 0, "views": 0}
    for it in range(n_cases):
        G = int(rng.integers(1, 13))
        lens = [int(rng.choice([1160, 1180, 1200, 1220, 1240, 1240, 1200])) for _ in range(G)]
        for g in range(G):  # some groups that are not whole lanes (they take the general kernel in the same call)
            if rng.random() < 0.25:
                lens[g] = int(rng.integers(900, 1280))
        T = sum(lens)
        gid = np.repeat(np.arange(G), lens).astype(np.int32)
        if rng.random() < 0.5:
            gid = gid[rng.permutation(T)]  # members of a group anywhere in time
        gid_p = gid if rng.random() < 0.6 else gid[rng.permutation(T)]  # same lengths, other positions
        C = 2 * int(rng.integers(4, 40))
        view = rng.random() < 0.4
        c0 = 2 * int(rng.integers(0, 6)) if view else 0
        Ct = C + c0 + (2 * int(rng.integers(0, 6)) if view else 0)
        style = rng.choice(["cont", "cont", "dyadic", "near", "const_stretch"])
        f = lambda: 12 + 6 * rng.standard_normal((T, Ct))  # noqa: E731
        full = {"X": f(), "y": f() + 3, "Xp": f()}
        if style == "dyadic":
            full = {k: np.round(v * 4) / 4 for k, v in full.items()}
        sl = slice(c0, c0 + C)
        X, y, Xp = (full[k][:, sl] for k in ("X", "y", "Xp"))
        if style == "near":
            for c in rng.choice(C, size=min(C, 3), replace=False):
                t = rng.choice(T, size=40, replace=False)
                Xp[t[:20], c] = Xp[t[20:], c] * (1 + rng.choice([0, 1e-16, 2e-16, -1e-16], size=20))
        if style == "const_stretch":
            for c in rng.choice(C, size=min(C, 3), replace=False):
                t0_ = int(rng.integers(0, T - 60))
                Xp[t0_:t0_ + 40, c] = Xp[t0_, c]
        if rng.random() < 0.3:
            X[0, int(rng.integers(0, C))] = np.nan
        if rng.random() < 0.3:
            Xp[int(rng.integers(0, T)), int(rng.integers(0, C))] = np.inf
        if rng.random() < 0.3:
            y[int(rng.integers(0, T)), int(rng.integers(0, C))] = np.nan
        dev = {k: ctx.to_device(v) for k, v in full.items()}
        res = {}
        for name, env in (("dma", {}), ("late", {"SD_FD_LATE": "1"}), ("regs", {"SD_FX_NODMA": "1"})):
            os.environ.update(env)
            big = ctx.to_device(np.full((T, Ct), -777.0))
            ctx.prof_reset()
            ctx.prof_enable(True)
            _, st = ctx.bcsd_fit_predict(0, dev["X"].cells(c0, c0 + C), dev["y"].cells(c0, c0 + C), gid, G, dev["Xp"].cells(c0, c0 + C), gid_p,
                                         out=big.cells(c0, c0 + C))
            ctx.prof_enable(False)
            if name == "dma" and "bcsd_fd_kernel" in ctx.prof():
                stats["dma_launches"] += 1
            res[name] = (big.to_host(), st)
            for k in env:
                del os.environ[k]
            big.free()
        for d in dev.values():
            d.free()
        ref, st_ref = res["regs"]
        for name in ("dma", "late"):
            got, st = res[name]
            assert np.array_equal(st, st_ref), (it, name, "status")
            assert np.array_equal(got, ref, equal_nan=True), (it, name, style, G, lens, C, c0, Ct)
        n = min(C, 4)
        exp, est = bo.pointwise_fit_predict(0, X[:, :n].copy(), y[:, :n].copy(), Xp[:, :n].copy(), gid, gid_p, G=G)
        assert np.array_equal(st_ref[:n], est), (it, "oracle status")
        ok = est == 0
        if ok.any():
            assert_close(ref[:, sl][:, :n][:, ok], exp[:, ok], what=f"case {it} {style}")
        stats["cases"] += 1
        stats["views"] += int(view)
        stats["worklist_styles"] += int(style in ("dyadic", "near", "const_stretch"))
    print(f"fuzz_fd: {stats} in {time.time() - t0:.0f} s, seed {seed}: all equal to the register-tile kernel bit for bit, oracle within 1e-6")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
