"""Thin object layer over the C ABI: contexts, HBM-resident fields and fitted-state handles."""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from ._lib import check, ptr


class DeviceArray:
    """A float64/int field resident in HBM (owned by the library's allocator, not torch)."""

    def __init__(self, ctx, shape, dtype=np.float64, dptr=None, owner=True, ld=None, base=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ld = self.shape[-1] if ld is None else int(ld)  # elements between consecutive rows (>= cells)
        self._base = base  # keeps the parent of a view alive
        self._owner = owner
        if dptr is None:
            p = C.c_void_p()
            check(ctx.lib.sd_dev_alloc(ctx.handle, self.nbytes, C.byref(p)))
            dptr = p.value
        self.ptr = dptr
        if owner:
            self._fin = weakref.finalize(self, DeviceArray._free, ctx, dptr)

    @staticmethod
    def _free(ctx, dptr):
        if ctx.handle is not None:
            ctx.lib.sd_dev_free(ctx.handle, C.c_void_p(dptr))

    def free(self):
        if self._owner and self._fin.alive:
            self._fin()

    @property
    def vptr(self):
        return C.c_void_p(self.ptr)

    @property
    def base(self):
        """the array this view was cut from (None for an array that owns its memory)"""
        return self._base

    def cells(self, c0, c1):
        """View of the cell range [c0, c1) of a [..., C] field: same rows, leading dimension of the parent
        (how a cell shard of a resident grid is handed to the engine without a copy)."""
        assert 0 <= c0 < c1 <= self.shape[-1]
        return DeviceArray(self.ctx, self.shape[:-1] + (c1 - c0,), self.dtype, dptr=self.ptr + c0 * self.dtype.itemsize,
                           owner=False, ld=self.ld, base=self)

    def to_host(self):
        if self.ld != self.shape[-1]:  # strided view: copy the parent's rows and slice on the host
            rows = int(np.prod(self.shape[:-1], dtype=np.int64))
            nbytes = ((rows - 1) * self.ld + self.shape[-1]) * self.dtype.itemsize
            flat = np.empty(nbytes // self.dtype.itemsize, dtype=self.dtype)
            check(self.ctx.lib.sd_memcpy_d2h(self.ctx.handle, ptr(flat), self.vptr, nbytes))
            full = np.zeros(rows * self.ld, dtype=self.dtype)
            full[:flat.size] = flat
            return full.reshape(rows, self.ld)[:, :self.shape[-1]].reshape(self.shape).copy()
        out = np.empty(self.shape, dtype=self.dtype)
        check(self.ctx.lib.sd_memcpy_d2h(self.ctx.handle, ptr(out), self.vptr, self.nbytes))
        return out

    def copy_from_host(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.shape == self.shape, (a.shape, self.shape)
        check(self.ctx.lib.sd_memcpy_h2d(self.ctx.handle, self.vptr, ptr(a), self.nbytes))
        return self


class _State:
    def __init__(self, ctx, handle, destroy):
        self.ctx = ctx
        self.handle = handle
        self._fin = weakref.finalize(self, _State._destroy, ctx, handle, destroy)

    @staticmethod
    def _destroy(ctx, handle, destroy):
        if ctx.handle is not None and handle:
            destroy(C.c_void_p(handle))

    def close(self):
        if self._fin.alive:
            self._fin()
        self.handle = None

    @property
    def vptr(self):
        if self.handle is None:
            raise ValueError("state has been destroyed")
        return C.c_void_p(self.handle)


class BcsdState(_State):
    def info(self):
        kind, G, ra = C.c_int(), C.c_int(), C.c_int()
        T, Cc = C.c_int64(), C.c_int64()
        check(self.ctx.lib.sd_bcsd_state_info(self.vptr, C.byref(kind), C.byref(G), C.byref(T), C.byref(Cc), C.byref(ra)))
        return dict(kind=kind.value, G=G.value, T=T.value, C=Cc.value, return_anoms=bool(ra.value & _lib.BCSD_RETURN_ANOMS),
                    detrend=bool(ra.value & _lib.BCSD_QM_DETREND))

    def status(self):
        st = np.empty(self.info()["C"], dtype=np.int32)
        check(self.ctx.lib.sd_bcsd_state_status(self.vptr, ptr(st)))
        return st

    def set_tails(self, extrapolate="both", n_endpoints=10):
        """CunnaneTransformer(extrapolate, n_endpoints) of the fitted inverse CDFs (quantile.py:418-431, 523-545) for later
        predict calls: which tails continue along the least-squares line through the first / last ``n_endpoints`` points."""
        masks = {"both": 3, "min": _lib.QT_TAIL_LOWER, "max": _lib.QT_TAIL_UPPER, "1to1": 0, None: 0}
        if extrapolate not in masks:
            raise ValueError(f"extrapolate={extrapolate!r}: expected one of 'min', 'max', 'both', '1to1', None")
        check(self.ctx.lib.sd_bcsd_state_set_tails(self.vptr, masks[extrapolate], int(n_endpoints)))

    def export(self):
        i = self.info()
        ys = np.empty((i["C"], i["T"]))
        xc = np.empty((i["C"], i["G"]))
        yc = np.empty((i["C"], i["G"]))
        st = np.empty(i["C"], dtype=np.int32)
        off = np.empty(i["G"] + 1, dtype=np.int64)
        check(self.ctx.lib.sd_bcsd_state_export(self.vptr, ptr(ys), ptr(xc), ptr(yc), ptr(st), ptr(off)))
        e = dict(info=i, y_sorted=ys, x_climo=xc, y_climo=yc, status=st, group_offsets=off)
        if i["detrend"]:  # slope, intercept of every fitted segment's line (quantile.py:97)
            e["y_trend"] = np.empty((i["C"], i["G"], 2))
            check(self.ctx.lib.sd_bcsd_state_get_trend(self.vptr, ptr(e["y_trend"])))
        return e


class AnalogState(_State):
    def info(self):
        T, Cc, F = C.c_int64(), C.c_int64(), C.c_int()
        check(self.ctx.lib.sd_analog_state_info(self.vptr, C.byref(T), C.byref(F), C.byref(Cc)))
        return dict(T=T.value, F=F.value, C=Cc.value)


class LinregState(_State):
    def info(self):
        T, Cc, F = C.c_int64(), C.c_int64(), C.c_int()
        check(self.ctx.lib.sd_linreg_state_info(self.vptr, C.byref(T), C.byref(F), C.byref(Cc)))
        return dict(T=T.value, F=F.value, C=Cc.value)

    def export(self):
        """coef [F, C], intercept [C], fit_error [C], status [C]; with a threshold also logistic_coef [F, C],
        logistic_intercept [C] and thresh_dropped [C] (cells whose samples all exceed: probability 1, gard.py:426-437)"""
        i = self.info()
        coef, icpt, err = np.empty((i["F"], i["C"])), np.empty(i["C"]), np.empty(i["C"])
        logit = np.full((i["F"] + 1, i["C"]), np.nan)
        dropped = np.full(i["C"], -1, dtype=np.int32)
        status = np.empty(i["C"], dtype=np.int32)
        check(self.ctx.lib.sd_linreg_state_export(self.vptr, ptr(coef), ptr(icpt), ptr(err), ptr(logit), ptr(dropped), ptr(status)))
        out = dict(coef=coef, intercept=icpt, fit_error=err, status=status, T=i["T"])
        if (dropped >= 0).all():  # a model with a threshold
            out.update(logistic_coef=logit[:-1], logistic_intercept=logit[-1], thresh_dropped=dropped.astype(bool))
        return out


class QmState(_State):
    def info(self):
        T, Cc = C.c_int64(), C.c_int64()
        check(self.ctx.lib.sd_qm_state_info(self.vptr, C.byref(T), C.byref(Cc)))
        return dict(T=T.value, C=Cc.value)

    def export(self, with_y=True):
        i = self.info()
        xs = np.empty((i["C"], i["T"]))
        ys = np.empty((i["C"], i["T"])) if with_y else None
        status = np.empty(i["C"], dtype=np.int32)
        check(self.ctx.lib.sd_qm_state_export(self.vptr, ptr(xs), None if ys is None else ptr(ys), ptr(status)))
        return dict(x_sorted=xs, y_sorted=ys, status=status)


class Context:
    """One GPU + one HIP stream.  Calls on a context are serialised."""

    def __init__(self, device=0, lib_path=None):
        self.lib = _lib.load(lib_path)  # lib_path: development library with A/B switches (tests / tools only)
        h = C.c_void_p()
        check(self.lib.sd_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)

    def close(self):
        if self.handle is not None:
            self.lib.sd_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- info / timing ----
    def device_info(self):
        name = C.create_string_buffer(256)
        cu, mem = C.c_int(), C.c_int64()
        check(self.lib.sd_ctx_device_info(self.handle, name, 256, C.byref(cu), C.byref(mem)))
        return dict(name=name.value.decode(), compute_units=cu.value, hbm_bytes=mem.value)

    def synchronize(self):
        check(self.lib.sd_ctx_synchronize(self.handle))

    def release_cached(self):
        """Give the context's cached device blocks (recycled state / scratch buffers) back to the driver."""
        check(self.lib.sd_ctx_release_cached(self.handle))

    def timer_start(self):
        check(self.lib.sd_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.sd_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        check(self.lib.sd_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self):
        check(self.lib.sd_prof_reset(self.handle))

    def prof(self):
        buf = C.create_string_buffer(4096)
        check(self.lib.sd_prof_names(self.handle, buf, 4096))
        out = {}
        for name in filter(None, buf.value.decode().split(";")):
            ms, n = C.c_double(), C.c_int64()
            check(self.lib.sd_prof_query(self.handle, name.encode(), C.byref(ms), C.byref(n)))
            out[name] = dict(ms=ms.value, launches=n.value)
        return out

    # ---- memory ----
    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def to_device(self, a, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype)
        return DeviceArray(self, a.shape, dtype).copy_from_host(a)

    # ---- float32 transport: a float32 host field crosses PCIe as float32 and is widened on the device (exact), a result
    # wanted as float32 is narrowed there (round to nearest, what ``.astype(np.float32)`` does on the host) ----
    @staticmethod
    def _is_f32_host(a):
        return isinstance(a, np.ndarray) and a.dtype == np.float32

    def widen_to_device(self, a32):
        """float32 host array -> float64 DeviceArray of the same shape (half the host-to-device bytes of an upcast on the host)"""
        a32 = np.ascontiguousarray(a32, dtype=np.float32)
        d32 = DeviceArray(self, a32.shape, np.float32).copy_from_host(a32)
        d64 = DeviceArray(self, a32.shape, np.float64)
        check(self.lib.sd_convert_f32_to_f64_dev(self.handle, d32.vptr, a32.size, d64.vptr))
        self.synchronize()
        d32.free()
        return d64

    def narrow_to_host(self, d64, out=None):
        """float64 DeviceArray (contiguous) -> float32 host array: narrowed on the device, half the device-to-host bytes"""
        if d64.ld != d64.shape[-1]:
            raise ValueError("narrow_to_host needs a contiguous DeviceArray")
        n = int(np.prod(d64.shape, dtype=np.int64))
        d32 = DeviceArray(self, d64.shape, np.float32)
        check(self.lib.sd_convert_f64_to_f32_dev(self.handle, d64.vptr, n, d32.vptr))
        res = np.empty(d64.shape, dtype=np.float32) if out is None else out
        check(self.lib.sd_memcpy_d2h(self.handle, ptr(res), d32.vptr, res.nbytes))
        d32.free()
        return res

    def wrap(self, dptr, shape, dtype=np.float64):
        """View foreign device memory (e.g. a torch tensor's data_ptr()) without owning it."""
        return DeviceArray(self, shape, dtype, dptr=int(dptr), owner=False)

    def synth_fill(self, out, kind, seed, stream, c_offset=0, c_full=None, base=None, amp=1.0, cell_scale=0.0,
                   p_dry=0.0, stream2=None, amp2=0.0):
        T, Cc = out.shape
        c_full = Cc if c_full is None else c_full
        b = None if base is None else np.ascontiguousarray(base, dtype=np.float64)
        check(self.lib.sd_synth_fill(self.handle, out.vptr, T, Cc, out.ld, c_offset, c_full, kind, seed, stream, ptr(b), amp,
                                     cell_scale, p_dry, -1 if stream2 is None else stream2, amp2))
        return out

    # ---- BCSD ----
    @staticmethod
    def _field2(name, a, T=None, Cc=None):
        """[T, C] field (numpy or DeviceArray) with the expected sizes: the C ABI takes raw pointers and cannot check them."""
        if a is None:
            return None
        if not isinstance(a, DeviceArray):
            a = _lib.as_f64(a)
        if len(a.shape) != 2 or (T is not None and a.shape[0] != T) or (Cc is not None and a.shape[1] != Cc):
            raise ValueError(f"{name}: expected a [{'T' if T is None else T}, {'C' if Cc is None else Cc}] field, got shape {tuple(a.shape)}")
        return a

    @staticmethod
    def _group_ids(name, gid, T, G):
        gid = _lib.as_i32(gid)
        if gid.shape != (T,):
            raise ValueError(f"{name}: expected {T} group ids, got shape {gid.shape}")
        if T and (gid.min() < 0 or gid.max() >= G):
            raise ValueError(f"{name}: group ids must lie in [0, {G})")
        return gid

    @staticmethod
    def _bcsd_options(return_anoms, detrend):
        return (_lib.BCSD_RETURN_ANOMS if return_anoms else 0) | (_lib.BCSD_QM_DETREND if detrend else 0)

    def bcsd_fit(self, kind, X, y, gid, G, return_anoms=True, detrend=False):
        """X, y: numpy [T,C] (host path) or DeviceArray [T,C] (resident path); X may be None for PR.  ``detrend``:
        qm_kwargs={'detrend': True} (quantile.py:95-98,128-145)."""
        return_anoms = self._bcsd_options(return_anoms, detrend)
        if self._is_f32_host(y) and (X is None or self._is_f32_host(X)) and y.ndim == 2 and y.size > 0:
            y = self.widen_to_device(y)  # float32 grids: 4 bytes per sample over PCIe, widened in HBM
            X = None if X is None else self.widen_to_device(X)
        y = self._field2("y", y)
        X = self._field2("X", X, *y.shape)
        if X is not None and isinstance(X, DeviceArray) != isinstance(y, DeviceArray):
            raise ValueError("X and y must both be host arrays or both be DeviceArrays")
        gid = self._group_ids("group_id", gid, y.shape[0], G)
        h = C.c_void_p()
        if isinstance(y, DeviceArray):
            T, Cc = y.shape
            assert X is None or X.ld == y.ld
            check(self.lib.sd_bcsd_fit_dev(self.handle, kind, None if X is None else X.vptr, y.vptr, y.ld, ptr(gid), G, T, Cc,
                                           int(return_anoms), C.byref(h)))
        else:
            y = _lib.as_f64(y)
            X = None if X is None else _lib.as_f64(X)
            T, Cc = y.shape
            check(self.lib.sd_bcsd_fit(self.handle, kind, ptr(X), ptr(y), ptr(gid), G, T, Cc, int(return_anoms), C.byref(h)))
        return BcsdState(self, h.value, self.lib.sd_bcsd_state_destroy)

    def bcsd_fit_groups(self, kind, X, y, order, offsets, return_anoms=True, detrend=False):
        """Fit on explicitly listed (possibly overlapping) groups: ``order`` = time indices group by group,
        ``offsets[G+1]`` (time_grouper='daily_nasa-nex': bcsd.py:36-38,50-55)."""
        return_anoms = self._bcsd_options(return_anoms, detrend)
        y = self._field2("y", y)
        X = self._field2("X", X, *y.shape)
        T, Cc = y.shape
        order = _lib.as_i32(order)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        G = len(offsets) - 1
        if G < 1 or offsets[0] != 0 or offsets[-1] != len(order) or (np.diff(offsets) < 0).any():
            raise ValueError("group offsets must start at 0, not decrease and end at len(order)")
        if len(order) and (order.min() < 0 or order.max() >= T):
            raise ValueError(f"group order entries must lie in [0, {T})")
        h = C.c_void_p()
        if isinstance(y, DeviceArray):
            assert X is None or X.ld == y.ld
            check(self.lib.sd_bcsd_fit_groups_dev(self.handle, kind, None if X is None else X.vptr, y.vptr, y.ld, ptr(order),
                                                  ptr(offsets), G, T, Cc, int(return_anoms), C.byref(h)))
        else:
            check(self.lib.sd_bcsd_fit_groups(self.handle, kind, ptr(X), ptr(y), ptr(order), ptr(offsets), G, T, Cc,
                                              int(return_anoms), C.byref(h)))
        return BcsdState(self, h.value, self.lib.sd_bcsd_state_destroy)

    def _result_buffer(self, out, shape, on_device):
        """the caller's result buffer, checked (a wrong one would be written past its end), or a fresh one"""
        if out is None:
            return self.empty(shape) if on_device else np.empty(shape)
        if on_device:
            if not isinstance(out, DeviceArray) or tuple(out.shape) != tuple(shape) or out.dtype != np.float64 or out.ld < shape[-1]:
                raise ValueError(f"out: expected a float64 DeviceArray of shape {tuple(shape)}")
        elif not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == tuple(shape) and out.flags["C_CONTIGUOUS"]):
            raise ValueError(f"out: expected a C-contiguous float64 array of shape {tuple(shape)}")
        return out

    def bcsd_predict_trend(self, state, Xp, gid_p, gid_trend, G_trend, out=None):
        """Predict with a climate-trend grouper of its own: rolling mean over ``gid_trend`` groups, climatologies and
        quantile mapping over ``gid_p`` (groups of the state) -- bcsd.py:247-267."""
        info = state.info()
        Xp = self._field2("X", Xp, None, info["C"])
        Tp, Cc = Xp.shape
        gid_p = self._group_ids("group_id", gid_p, Tp, info["G"])
        gid_trend = self._group_ids("trend_group_id", gid_trend, Tp, G_trend)
        status = np.empty(Cc, dtype=np.int32)
        out = self._result_buffer(out, (Tp, Cc), isinstance(Xp, DeviceArray))
        if isinstance(Xp, DeviceArray):
            check(self.lib.sd_bcsd_predict_trend_dev(self.handle, state.vptr, Xp.vptr, Xp.ld, ptr(gid_p), ptr(gid_trend), G_trend, Tp,
                                                     out.vptr, out.ld, ptr(status)))
        else:
            check(self.lib.sd_bcsd_predict_trend(self.handle, state.vptr, ptr(Xp), ptr(gid_p), ptr(gid_trend), G_trend, Tp, ptr(out),
                                                 ptr(status)))
        return out, status

    def bcsd_predict(self, state, Xp, gid_p, out=None, out_dtype=None):
        """``out_dtype=np.float32`` with a float32 host ``Xp``: the field goes in and the result comes back as float32
        (widened / narrowed on the device); everything else as before (float64 results)."""
        info = state.info()
        want32 = out_dtype is not None and np.dtype(out_dtype) == np.float32
        if out_dtype is not None and not want32 and np.dtype(out_dtype) != np.float64:
            raise ValueError("bcsd_predict: out_dtype must be float64 or float32")
        if self._is_f32_host(Xp) and Xp.ndim == 2 and Xp.size > 0 and out is None:
            d_out, status = self.bcsd_predict(state, self.widen_to_device(Xp), gid_p)
            return (self.narrow_to_host(d_out) if want32 else d_out.to_host()), status
        if want32:  # (float64 or resident input: the result is narrowed where it is)
            if out is not None:
                raise ValueError("bcsd_predict: out_dtype=float32 does not combine with a float64 `out` buffer")
            res, status = self.bcsd_predict(state, Xp, gid_p)
            return (self.narrow_to_host(res) if isinstance(res, DeviceArray) else res.astype(np.float32)), status
        Xp = self._field2("X", Xp, None, info["C"])
        gid_p = self._group_ids("group_id", gid_p, Xp.shape[0], info["G"])
        Cc = info["C"]
        status = np.empty(Cc, dtype=np.int32)
        if isinstance(Xp, DeviceArray):
            Tp = Xp.shape[0]
            out = self._result_buffer(out, (Tp, Cc), True)
            check(self.lib.sd_bcsd_predict_dev(self.handle, state.vptr, Xp.vptr, Xp.ld, ptr(gid_p), Tp, out.vptr, out.ld, ptr(status)))
        else:
            Xp = _lib.as_f64(Xp)
            Tp = Xp.shape[0]
            out = self._result_buffer(out, (Tp, Cc), False)  # (a reused result buffer)
            check(self.lib.sd_bcsd_predict(self.handle, state.vptr, ptr(Xp), ptr(gid_p), Tp, ptr(out), ptr(status)))
        return out, status

    def bcsd_fit_predict(self, kind, X, y, gid, G, Xp, gid_p, return_anoms=True, out=None, detrend=False):
        """Fused resident path (DeviceArrays only)."""
        return_anoms = self._bcsd_options(return_anoms, detrend)
        y = self._field2("y", y)
        X = self._field2("X", X, *y.shape)
        Xp = self._field2("Xp", Xp, None, y.shape[1])
        T, Cc = y.shape
        Tp = Xp.shape[0]
        gid, gid_p = self._group_ids("group_id", gid, T, G), self._group_ids("group_id_p", gid_p, Tp, G)
        out = self.empty((Tp, Cc)) if out is None else out
        if tuple(out.shape) != (Tp, Cc):
            raise ValueError(f"out: expected shape {(Tp, Cc)}, got {tuple(out.shape)}")
        status = np.empty(Cc, dtype=np.int32)
        assert X is None or X.ld == y.ld
        check(self.lib.sd_bcsd_fit_predict_dev(self.handle, kind, None if X is None else X.vptr, y.vptr, y.ld, ptr(gid), G, T,
                                               Cc, int(return_anoms), Xp.vptr, Xp.ld, ptr(gid_p), Tp, out.vptr, out.ld,
                                               ptr(status)))
        return out, status

    def bcsd_import(self, exported):
        i = exported["info"]
        h = C.c_void_p()
        check(self.lib.sd_bcsd_state_import(
            self.handle, i["kind"], i["G"], i["T"], i["C"], self._bcsd_options(i["return_anoms"], i.get("detrend", False)),
            ptr(_lib.as_f64(exported["y_sorted"])),
            ptr(_lib.as_f64(exported["x_climo"])), ptr(_lib.as_f64(exported["y_climo"])), ptr(_lib.as_i32(exported["status"])),
            ptr(np.ascontiguousarray(exported["group_offsets"], dtype=np.int64)), C.byref(h)))
        st = BcsdState(self, h.value, self.lib.sd_bcsd_state_destroy)
        if i.get("detrend", False):
            trend = _lib.as_f64(exported["y_trend"])
            if trend.shape != (i["C"], i["G"], 2):
                raise ValueError(f"y_trend: expected shape {(i['C'], i['G'], 2)}, got {trend.shape}")
            check(self.lib.sd_bcsd_state_set_trend(st.vptr, ptr(trend)))
        return st

    # ---- quantile-mapping regressors ----
    def qm_fit(self, X, y=None):
        """X, y [T, C] numpy or DeviceArray -> QmState (sorted series per cell); y=None keeps only the CDF of X
        (CunnaneTransformer)."""
        X = self._field2("X", X)
        y = self._field2("y", y, *X.shape)
        if y is not None and isinstance(X, DeviceArray) != isinstance(y, DeviceArray):
            raise ValueError("X and y must both be host arrays or both be DeviceArrays")
        h = C.c_void_p()
        if isinstance(X, DeviceArray):
            T, Cc = X.shape
            assert y is None or X.ld == y.ld
            check(self.lib.sd_qm_fit_dev(self.handle, X.vptr, None if y is None else y.vptr, X.ld, T, Cc, C.byref(h)))
        else:
            X = _lib.as_f64(X)
            y = None if y is None else _lib.as_f64(y)
            T, Cc = X.shape
            check(self.lib.sd_qm_fit(self.handle, ptr(X), None if y is None else ptr(y), T, Cc, C.byref(h)))
        return QmState(self, h.value, self.lib.sd_qm_state_destroy)

    def qm_cunnane(self, state, direction, X, extrapolate="both", n_endpoints=10, out=None):
        """CunnaneTransformer.transform (direction 0) / inverse_transform (1) of X [Tp, C] on the fitted CDFs."""
        Cc = state.info()["C"]
        X = self._field2("X", X, None, Cc)
        status = np.empty(Cc, dtype=np.int32)
        if extrapolate not in _lib.EXTRAP_CODES:
            raise ValueError(f"unknown value for extrapolate: {extrapolate}")
        code = _lib.EXTRAP_CODES[extrapolate]
        if isinstance(X, DeviceArray):
            Tp = X.shape[0]
            out = self.empty((Tp, Cc)) if out is None else out
            check(self.lib.sd_qm_cunnane_dev(self.handle, state.vptr, int(direction), code, int(n_endpoints), X.vptr, X.ld, Tp,
                                             out.vptr, out.ld, ptr(status)))
        else:
            X = _lib.as_f64(X)
            Tp = X.shape[0]
            out = np.empty((Tp, Cc))
            check(self.lib.sd_qm_cunnane(self.handle, state.vptr, int(direction), code, int(n_endpoints), ptr(X), Tp, ptr(out),
                                         ptr(status)))
        return out, status

    def qm_predict(self, state, model, Xp, extrapolate=None, n_endpoints=10, out=None):
        """extrapolate: None, 'min', 'max', 'both' or '1to1' (``True`` is accepted for '1to1')."""
        extrapolate = "1to1" if extrapolate is True else (None if extrapolate is False else extrapolate)
        if extrapolate not in _lib.QM_EXTRAP_CODES:
            raise ValueError(f"unknown value for extrapolate: {extrapolate}")
        code = _lib.QM_EXTRAP_CODES[extrapolate]
        Cc = state.info()["C"]
        Xp = self._field2("X", Xp, None, Cc)
        status = np.empty(Cc, dtype=np.int32)
        Tp = Xp.shape[0]
        if isinstance(Xp, DeviceArray):
            out = self.empty((Tp, Cc)) if out is None else out
            check(self.lib.sd_qm_predict_dev(self.handle, state.vptr, int(model), code, int(n_endpoints), Xp.vptr, Xp.ld, Tp,
                                             out.vptr, out.ld, ptr(status)))
        else:
            out = np.empty((Tp, Cc))
            check(self.lib.sd_qm_predict(self.handle, state.vptr, int(model), code, int(n_endpoints), ptr(Xp), Tp, ptr(out),
                                         ptr(status)))
        return out, status

    # ---- analogs ----
    def analog_fit(self, X, y):
        """X [T,F,C], y [T,C] numpy or DeviceArray."""
        if not isinstance(X, DeviceArray):
            X, y = _lib.as_f64(X), _lib.as_f64(y)
        if len(X.shape) != 3 or tuple(y.shape) != (X.shape[0], X.shape[2]) or isinstance(X, DeviceArray) != isinstance(y, DeviceArray):
            raise ValueError(f"expected X [T, F, C] and y [T, C] of the same kind, got {tuple(X.shape)} and {tuple(y.shape)}")
        h = C.c_void_p()
        if isinstance(X, DeviceArray):
            T, F, Cc = X.shape
            assert X.ld == y.ld
            check(self.lib.sd_analog_fit_dev(self.handle, X.vptr, y.vptr, y.ld, T, F, Cc, C.byref(h)))
        else:
            X, y = _lib.as_f64(X), _lib.as_f64(y)
            T, F, Cc = X.shape
            check(self.lib.sd_analog_fit(self.handle, ptr(X), ptr(y), T, F, Cc, C.byref(h)))
        return AnalogState(self, h.value, self.lib.sd_analog_state_destroy)

    def analog_fit_predict(self, X, y, Xq, k, kind, thresh=None, out=None):
        """AnalogBase.fit + PureAnalog.predict in one call without a fitted state (sd_analog_fit_predict*): X [T,F,C], y [T,C],
        Xq [Tq,F,C], all numpy or all DeviceArray -> (out [Tq,3,C], status [C]); bit-identical to analog_fit -> analog_predict."""
        dev = isinstance(X, DeviceArray)
        if not dev:
            X, y, Xq = _lib.as_f64(X), _lib.as_f64(y), _lib.as_f64(Xq)
        if isinstance(y, DeviceArray) != dev or isinstance(Xq, DeviceArray) != dev:
            raise ValueError("X, y and Xq must all be numpy arrays or all DeviceArrays")
        if len(X.shape) != 3 or tuple(y.shape) != (X.shape[0], X.shape[2]):
            raise ValueError(f"expected X [T, F, C] and y [T, C], got {tuple(X.shape)} and {tuple(y.shape)}")
        T, F, Cc = X.shape
        if len(Xq.shape) != 3 or Xq.shape[1] != F or Xq.shape[2] != Cc:
            raise ValueError(f"Xq: expected a [Tq, {F}, {Cc}] field, got shape {tuple(Xq.shape)}")
        k = int(k)
        if k < 1 or k > T:
            raise ValueError(f"k={k}: expected 1 <= k <= {T} (the number of training samples)")
        Tq = Xq.shape[0]
        status = np.empty(Cc, dtype=np.int32)
        has_t, tv = (0, 0.0) if thresh is None else (1, float(thresh))
        if dev:
            assert X.ld == y.ld
            out = self.empty((Tq, 3, Cc)) if out is None else out
            check(self.lib.sd_analog_fit_predict_dev(self.handle, X.vptr, y.vptr, y.ld, T, F, Cc, Xq.vptr, Xq.ld, Tq, k, kind, has_t, tv,
                                                     out.vptr, out.ld, ptr(status)))
        else:
            if out is None:
                out = np.empty((Tq, 3, Cc))
            elif not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == (Tq, 3, Cc) and out.flags.c_contiguous):
                raise ValueError(f"out: expected a C-contiguous float64 array of shape {(Tq, 3, Cc)}")
            check(self.lib.sd_analog_fit_predict(self.handle, ptr(X), ptr(y), T, F, Cc, ptr(Xq), Tq, k, kind, has_t, tv, ptr(out), ptr(status)))
        return out, status

    def analog_predict(self, state, Xq, k, kind, thresh=None, sample_inds=None, want_neighbors=False, out=None):
        info = state.info()
        Cc = info["C"]
        if not isinstance(Xq, DeviceArray):
            Xq = _lib.as_f64(Xq)
        if len(Xq.shape) != 3 or Xq.shape[1] != info["F"] or Xq.shape[2] != Cc:
            raise ValueError(f"Xq: expected a [Tq, {info['F']}, {Cc}] field, got shape {tuple(Xq.shape)}")
        k = int(k)
        if k < 1 or k > info["T"]:
            raise ValueError(f"k={k}: expected 1 <= k <= {info['T']} (the number of training samples)")
        if sample_inds is not None:
            if not isinstance(sample_inds, DeviceArray):
                sample_inds = _lib.as_i32(sample_inds)
            if tuple(sample_inds.shape) != (Xq.shape[0], Cc):
                raise ValueError(f"sample_inds: expected shape {(Xq.shape[0], Cc)}, got {tuple(sample_inds.shape)}")
            if not isinstance(sample_inds, DeviceArray) and len(sample_inds) and (np.min(sample_inds) < 0 or np.max(sample_inds) >= k):
                raise ValueError(f"sample_inds must lie in [0, {k})")
        status = np.empty(Cc, dtype=np.int32)
        has_t, tv = (0, 0.0) if thresh is None else (1, float(thresh))
        if isinstance(Xq, DeviceArray):
            Tq = Xq.shape[0]
            out = self.empty((Tq, 3, Cc)) if out is None else out
            samp = None if sample_inds is None else (sample_inds if isinstance(sample_inds, DeviceArray) else self.to_device(sample_inds, np.int32))
            inds = self.empty((Tq, k, Cc), np.int64) if want_neighbors else None
            dist = self.empty((Tq, k, Cc)) if want_neighbors else None
            assert not want_neighbors or out.ld == Cc
            check(self.lib.sd_analog_predict_dev(self.handle, state.vptr, Xq.vptr, Xq.ld, Tq, k, kind, has_t, tv,
                                                 None if samp is None else samp.vptr, out.vptr, out.ld,
                                                 None if inds is None else inds.vptr, None if dist is None else dist.vptr, ptr(status)))
        else:
            Xq = _lib.as_f64(Xq)
            Tq = Xq.shape[0]
            out = np.empty((Tq, 3, Cc))
            samp = None if sample_inds is None else _lib.as_i32(sample_inds)
            inds = np.empty((Tq, k, Cc), dtype=np.int64) if want_neighbors else None
            dist = np.empty((Tq, k, Cc)) if want_neighbors else None
            check(self.lib.sd_analog_predict(self.handle, state.vptr, ptr(Xq), Tq, k, kind, has_t, tv, ptr(samp), ptr(out),
                                             ptr(inds), ptr(dist), ptr(status)))
        return (out, status, inds, dist) if want_neighbors else (out, status)

    def analogreg_predict(self, state, Xq, k, thresh=None, out=None):
        """AnalogRegression.predict for every cell; ``thresh``: exceedance probability by logistic regression on the analogs
        (gard.py:201-212), linear model on the exceeding analogs."""
        info = state.info()
        Cc = info["C"]
        if not isinstance(Xq, DeviceArray):
            Xq = _lib.as_f64(Xq)
        if len(Xq.shape) != 3 or Xq.shape[1] != info["F"] or Xq.shape[2] != Cc:
            raise ValueError(f"Xq: expected a [Tq, {info['F']}, {Cc}] field, got shape {tuple(Xq.shape)}")
        status = np.empty(Cc, dtype=np.int32)
        has, thr = (0, 0.0) if thresh is None else (1, float(thresh))
        Tq = Xq.shape[0]
        if isinstance(Xq, DeviceArray):
            out = self.empty((Tq, 3, Cc)) if out is None else out
            check(self.lib.sd_analogreg_predict_dev(self.handle, state.vptr, Xq.vptr, Xq.ld, Tq, int(k), has, thr, out.vptr, out.ld,
                                                    ptr(status)))
        else:
            out = np.empty((Tq, 3, Cc))
            check(self.lib.sd_analogreg_predict(self.handle, state.vptr, ptr(Xq), Tq, int(k), has, thr, ptr(out), ptr(status)))
        return out, status

    # ---- PureRegression (thresh=None) ----
    def linreg_fit(self, X, y, thresh=None):
        """X [T,F,C], y [T,C] numpy or DeviceArray -> LinregState (coefficients, intercept, fit error per cell; with ``thresh``
        also the logistic model of the exceedance probability, gard.py:416-437)."""
        if not isinstance(X, DeviceArray):
            X, y = _lib.as_f64(X), _lib.as_f64(y)
        if len(X.shape) != 3 or tuple(y.shape) != (X.shape[0], X.shape[2]) or isinstance(X, DeviceArray) != isinstance(y, DeviceArray):
            raise ValueError(f"expected X [T, F, C] and y [T, C] of the same kind, got {tuple(X.shape)} and {tuple(y.shape)}")
        has, thr = (0, 0.0) if thresh is None else (1, float(thresh))
        h = C.c_void_p()
        T, F, Cc = X.shape
        if isinstance(X, DeviceArray):
            assert X.ld == y.ld
            check(self.lib.sd_linreg_fit_dev(self.handle, X.vptr, y.vptr, y.ld, T, F, Cc, has, thr, C.byref(h)))
        else:
            check(self.lib.sd_linreg_fit(self.handle, ptr(X), ptr(y), T, F, Cc, has, thr, C.byref(h)))
        return LinregState(self, h.value, self.lib.sd_linreg_state_destroy)

    def linreg_import(self, exported):
        """device state from ``LinregState.export()`` (pickling, checkpoint / resume)"""
        coef = _lib.as_f64(exported["coef"])
        F, Cc = coef.shape
        logit = dropped = None
        if "logistic_coef" in exported:
            logit = _lib.as_f64(np.vstack([exported["logistic_coef"], np.asarray(exported["logistic_intercept"]).reshape(1, Cc)]))
            dropped = _lib.as_i32(exported["thresh_dropped"])
        h = C.c_void_p()
        check(self.lib.sd_linreg_state_import(self.handle, int(exported["T"]), F, Cc, ptr(coef), ptr(_lib.as_f64(exported["intercept"])),
                                              ptr(_lib.as_f64(exported["fit_error"])), ptr(logit), ptr(dropped),
                                              ptr(_lib.as_i32(exported["status"])), C.byref(h)))
        return LinregState(self, h.value, self.lib.sd_linreg_state_destroy)

    def linreg_predict(self, state, Xq, out=None):
        info = state.info()
        Cc = info["C"]
        if not isinstance(Xq, DeviceArray):
            Xq = _lib.as_f64(Xq)
        if len(Xq.shape) != 3 or Xq.shape[1] != info["F"] or Xq.shape[2] != Cc:
            raise ValueError(f"Xq: expected a [Tq, {info['F']}, {Cc}] field, got shape {tuple(Xq.shape)}")
        status = np.empty(Cc, dtype=np.int32)
        if isinstance(Xq, DeviceArray):
            Tq = Xq.shape[0]
            out = self.empty((Tq, 3, Cc)) if out is None else out
            check(self.lib.sd_linreg_predict_dev(self.handle, state.vptr, Xq.vptr, Xq.ld, Tq, out.vptr, out.ld, ptr(status)))
        else:
            Xq = _lib.as_f64(Xq)
            Tq = Xq.shape[0]
            out = np.empty((Tq, 3, Cc))
            check(self.lib.sd_linreg_predict(self.handle, state.vptr, ptr(Xq), Tq, ptr(out), ptr(status)))
        return out, status


_default_ctx = None


def default_context():
    """Process-wide context on ``$SD_DEVICE`` / ``$LOCAL_RANK`` / device 0."""
    global _default_ctx
    if _default_ctx is None or _default_ctx.handle is None:
        dev = int(os.environ.get("SD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
    return _default_ctx
