"""Grid driver: ``PointWiseDownscaler`` with the reference's surface (core.py:200-448).

The reference loops over cells in Python (core.py:86-96 and 137-141), deep-copying the estimator
and building a pandas DataFrame per cell.  Here BCSD / analog estimators are fitted and applied to
*all cells in one batched launch* of the HIP engine; any other estimator (sklearn Pipelines ...)
takes the reference's per-cell loop unchanged in behaviour.

Inputs may be ``xarray.DataArray`` / ``xarray.Dataset`` (lazy import; xarray is optional) or the
tiny labelled-array stand-in ``GridArray`` (numpy values + dims + coords) used where xarray is
not installed.  Output type follows the input type.
"""
from __future__ import annotations

import copy

import numpy as np
import pandas as pd

from . import _lib
from .bcsd import BcsdBase, check_supported
from .gard import AnalogGridModel, AnalogRegression, PureAnalog, PureRegression, RegressionGridModel
from .quantile import (CunnaneGridModel, CunnaneTransformer, QmGridModel, QuantileMapper, QuantileMapperGridModel,
                       QuantileMappingReressor, check_extrapolate)

DEFAULT_FEATURE_DIM = "variable"


class GridArray:
    """Minimal labelled array: ``values`` (ndarray), ``dims`` (tuple of names), ``coords`` (dict)."""

    def __init__(self, values, dims, coords=None, name=None):
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        if self.values.ndim != len(self.dims):
            raise ValueError(f"values has {self.values.ndim} dims but dims={self.dims}")
        self.coords = dict(coords or {})
        self.name = name

    @property
    def shape(self):
        return self.values.shape

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def sizes(self):
        return dict(zip(self.dims, self.values.shape))

    # chunk structure like a dask-backed xarray object: PointWiseDownscaler walks the spatial blocks (core.py:256-262,
    # 300-336: xr.map_blocks); the values themselves stay plain NumPy
    chunksizes = None

    @property
    def chunks(self):
        return None if self.chunksizes is None else tuple(self.chunksizes[d] for d in self.dims)

    def chunk(self, chunks):
        """``chunks``: dim -> block length (or -1 for one block), like ``xarray.DataArray.chunk``"""
        g = GridArray(self.values, self.dims, self.coords, self.name)
        cs = {}
        for d, n in self.sizes.items():
            b = chunks.get(d, -1) if chunks else -1
            b = n if b in (-1, None) or b >= n else int(b)
            cs[d] = tuple([b] * (n // b) + ([n % b] if n % b else [])) if n else (0,)
        g.chunksizes = cs
        return g

    def isel(self, **indexers):
        """positional selection by slices along named dims (coords follow)"""
        key = tuple(indexers.get(d, slice(None)) for d in self.dims)
        coords = {}
        for k, v in self.coords.items():
            if k in indexers and np.ndim(v) == 1 and len(v) == self.sizes.get(k, -1):
                coords[k] = v[indexers[k]]
            else:
                coords[k] = v
        return GridArray(self.values[key], self.dims, coords, self.name)

    def transpose(self, *dims):
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            rest = [d for d in self.dims if d not in dims]
            dims = tuple(dims[:i]) + tuple(rest) + tuple(dims[i + 1:])
        perm = [self.dims.index(d) for d in dims]
        return GridArray(self.values.transpose(perm), dims, self.coords, self.name)

    def __repr__(self):
        return f"<GridArray {self.sizes}>"


class LazyGridArray(GridArray):
    """Result of predict / transform / inverse_transform on a spatially chunked GridArray: like the reference's
    ``xr.map_blocks`` result (core.py:256-262, 300-336) the blocks are computed when they are asked for.  ``iter_blocks()``
    yields ``(selection, GridArray)`` one spatial block at a time -- nothing but the current block is held --, ``values``
    assembles (and keeps) the whole field.  Only the first block is computed up front: it tells dims, dtype and the sizes of
    the non-spatial dims."""

    def __init__(self, thunks, first, spatial_dims, sizes, chunksizes, coords, name=None):
        self._thunks = list(thunks)  # [(selection, callable -> GridArray)]
        self._first = first
        self._full = None
        self.dims = tuple(first.dims)
        all_sizes = dict(first.sizes)
        all_sizes.update({d: int(sizes[d]) for d in spatial_dims})
        self._shape = tuple(all_sizes[d] for d in self.dims)
        self._dtype = first.dtype
        self.coords = dict(coords)
        self.name = name
        self.chunksizes = {d: (tuple(chunksizes[d]) if d in chunksizes else (all_sizes[d],)) for d in self.dims}

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return self._dtype

    @property
    def sizes(self):
        return dict(zip(self.dims, self._shape))

    @property
    def computed(self):
        return self._full is not None

    def iter_blocks(self):
        """(selection, GridArray) per spatial block.  Once ``values`` has assembled the field the blocks are views of it:
        nothing is computed a second time."""
        for i, (sel, thunk) in enumerate(self._thunks):
            if self._full is not None:
                idx = tuple(sel.get(d, slice(None)) for d in self.dims)
                coords = {k: (np.asarray(v)[sel[k]] if k in sel else v) for k, v in self.coords.items()}
                yield sel, GridArray(self._full[idx], self.dims, coords, self.name)
            elif i == 0 and self._first is not None:
                yield sel, self._first
            else:
                yield sel, thunk()

    @property
    def values(self):
        if self._full is None:
            with np.errstate(invalid="ignore"):
                full = np.full(self._shape, np.nan, dtype=self._dtype)
            for sel, block in self.iter_blocks():
                full[tuple(sel.get(d, slice(None)) for d in self.dims)] = block.values
            self._full, self._first = full, None
        return self._full

    def compute(self):
        return GridArray(self.values, self.dims, self.coords, self.name).chunk({d: self.chunksizes[d][0] for d in self.dims})

    def chunk(self, chunks):
        return self.compute().chunk(chunks)

    def isel(self, **indexers):
        return self.compute().isel(**indexers)

    def transpose(self, *dims):
        return self.compute().transpose(*dims)

    def __repr__(self):
        return f"<LazyGridArray {self.sizes} blocks={len(self._thunks)} computed={self.computed}>"


class GridDataset(dict):
    """Ordered mapping name -> GridArray (stand-in for xarray.Dataset)."""

    @property
    def chunksizes(self):
        first = next(iter(self.values()), None)
        return None if first is None else first.chunksizes

    @property
    def chunks(self):
        return self.chunksizes

    def chunk(self, chunks):
        return GridDataset({k: v.chunk(chunks) for k, v in self.items()})


def _is_xarray(obj):
    mod = type(obj).__module__
    return mod.startswith("xarray")


def _to_grid(obj, feature_dim):
    """-> (GridArray, was_xarray)"""
    if isinstance(obj, GridArray):
        return obj, False
    if isinstance(obj, GridDataset):
        names = list(obj)
        first = obj[names[0]]
        vals = np.stack([obj[n].values for n in names], axis=0)
        coords = dict(first.coords)
        coords[feature_dim] = np.array(names)
        return GridArray(vals, (feature_dim,) + first.dims, coords), False
    if _is_xarray(obj):
        import xarray as xr

        if isinstance(obj, xr.Dataset):
            obj = obj.to_array(feature_dim)  # core.py:429-430
        coords = {k: np.asarray(v.values) for k, v in obj.coords.items() if v.ndim <= 1}
        return GridArray(np.asarray(obj.values), obj.dims, coords, obj.name), True
    raise TypeError(f"unsupported input type {type(obj)}; expected xarray.DataArray/Dataset or GridArray")


def _from_grid(g, as_xarray):
    if not as_xarray:
        return g
    import xarray as xr

    coords = {k: ((k,), v) if np.ndim(v) == 1 and k in g.dims else v for k, v in g.coords.items()
              if (k in g.dims) or np.ndim(v) == 0}
    return xr.DataArray(g.values, dims=g.dims, coords=coords)


def _unchunked(obj):
    """a block handed to a child model: plain (loaded) data without chunk structure"""
    if isinstance(obj, GridDataset):
        return GridDataset({k: _unchunked(v) for k, v in obj.items()})
    if isinstance(obj, GridArray):
        return GridArray(obj.values, obj.dims, obj.coords, obj.name)
    if _is_xarray(obj):
        return obj.compute() if getattr(obj, "chunks", None) else obj
    return obj


def _time_index(g, dim):
    """core.py:52-64: pandas index of the time coordinate (RangeIndex when absent)."""
    if dim in g.coords:
        v = g.coords[dim]
        return v if isinstance(v, pd.Index) else pd.Index(np.asarray(v))
    return pd.RangeIndex(g.sizes[dim])


def _da_to_df(values_2d, index, columns):
    return pd.DataFrame(values_2d, columns=columns, index=index)


class _BatchedModels:
    """Fitted state of a whole grid held by the engine (replaces the object array of estimators)."""

    def __init__(self, kind, grid_model, mask, spatial_dims, spatial_shape, coords):
        self.kind = kind  # 'bcsd' | 'analog' | 'qm' | 'cunnane' | 'qmapper' | 'loop'
        self.grid_model = grid_model
        self.mask = mask
        self.spatial_dims = spatial_dims
        self.spatial_shape = spatial_shape
        self.coords = coords


class _BlockedModels:
    """Fitted state of a chunked grid: one fitted PointWiseDownscaler per spatial block (the reference maps
    ``_fit_wrapper`` over the blocks of a dask-backed input, core.py:256-262; the engine then batches the cells of a block)."""

    kind = "blocks"

    def __init__(self, spatial_dims, spatial_shape, chunksizes, blocks):
        self.spatial_dims = tuple(spatial_dims)
        self.spatial_shape = tuple(spatial_shape)
        self.chunksizes = chunksizes  # dim -> tuple of block lengths
        self.blocks = blocks          # list of ({dim: slice}, fitted child)

    @property
    def sizes(self):
        return dict(zip(self.spatial_dims, self.spatial_shape))


def _block_slices(dims, chunksizes):
    """all blocks of a chunked grid as {dim: slice} dicts, last dim fastest"""
    import itertools

    per_dim = []
    for d in dims:
        edges = np.concatenate([[0], np.cumsum(chunksizes[d])]).astype(int)
        per_dim.append([slice(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])])
    return [dict(zip(dims, combo)) for combo in itertools.product(*per_dim)]


def _isel(obj, sel):
    """positional block selection on GridArray / GridDataset / xarray objects, ignoring dims the object does not have"""
    if isinstance(obj, GridDataset):
        return GridDataset({k: _isel(v, sel) for k, v in obj.items()})
    return obj.isel(**{d: s for d, s in sel.items() if d in obj.dims})


class PointWiseDownscaler:
    """Pointwise downscaling model wrapper (core.py:200-448).

    Parameters
    ----------
    model : estimator implementing fit/predict
    dim : str, dimension to apply the model along (default ``time``)
    """

    def __init__(self, model, dim="time"):
        self._dim = dim
        self._model = model
        self._models = None
        if not hasattr(model, "fit"):
            raise TypeError(f"Type {type(model)} does not have the fit method required by PointWiseDownscaler")

    # ------------------------------------------------------------------------------------------
    def _to_feature_x(self, X, feature_dim=DEFAULT_FEATURE_DIM):
        """core.py:427-440 -> GridArray with dims (time, feature, *spatial)."""
        g, was_x = _to_grid(X, feature_dim)
        if feature_dim not in g.dims:
            vals = np.expand_dims(g.values, 1)
            dims = (g.dims[0], feature_dim) + g.dims[1:] if g.dims[0] == self._dim else None
            if dims is None:  # time is not leading: move it first, then insert the feature axis
                g = g.transpose(self._dim, ...)
                vals = np.expand_dims(g.values, 1)
                dims = (self._dim, feature_dim) + g.dims[1:]
            coords = dict(g.coords)
            coords[feature_dim] = np.array([f"{feature_dim}_0"])
            g = GridArray(vals, dims, coords, g.name)
        return g.transpose(self._dim, feature_dim, ...), was_x

    def _spatial_chunks(self, X, feature_dim):
        """(spatial dims, sizes, dim -> block lengths) of a chunked input, else None.  Time and feature dims are always taken
        whole (the reference needs them in one chunk: core.py:435-437 and the ``time: -1`` of its examples)."""
        if not getattr(X, "chunks", None):
            return None
        cs = X.chunksizes
        probe = X[list(X)[0]] if isinstance(X, GridDataset) else X
        if _is_xarray(X) and not hasattr(X, "dims"):
            return None
        dims = [d for d in probe.dims if d not in (self._dim, feature_dim)]
        sizes = dict(probe.sizes)
        chunksizes = {d: tuple(int(b) for b in cs[d]) if d in cs else (int(sizes[d]),) for d in dims}
        if all(len(chunksizes[d]) == 1 for d in dims):
            return None
        return dims, sizes, chunksizes

    def _apply_blocks(self, method, X, kwargs):
        """predict / transform / inverse_transform of a block-fitted grid: every block through its own fitted child, results
        assembled along the spatial dims (core.py:300-336 maps ``_predict_wrapper`` over the blocks)."""
        mdl = self._models
        was_x = _is_xarray(X)
        feature_dim = kwargs.get("feature_dim", DEFAULT_FEATURE_DIM)

        def run(sel, child):
            rg, _ = _to_grid(getattr(child, method)(_unchunked(_isel(X, sel)), **kwargs), feature_dim)
            return rg

        if not was_x:  # GridArray / GridDataset: blocks on demand (like the reference's map_blocks result)
            thunks = [(sel, (lambda s=sel, c=child: run(s, c))) for sel, child in mdl.blocks]
            first = thunks[0][1]()
            probe = X[list(X)[0]] if isinstance(X, GridDataset) else X
            coords = {k: v for k, v in first.coords.items() if k not in mdl.spatial_dims}
            for k in mdl.spatial_dims:
                if k in getattr(probe, "coords", {}):
                    coords[k] = probe.coords[k]
            return LazyGridArray(thunks, first, mdl.spatial_dims, mdl.sizes, mdl.chunksizes, coords)
        out = None
        for sel, child in mdl.blocks:
            rg = run(sel, child)
            if out is None:
                sizes = dict(rg.sizes)
                sizes.update(mdl.sizes)
                full = np.full([sizes[d] for d in rg.dims], np.nan, dtype=rg.dtype)
                coords = {k: v for k, v in rg.coords.items() if k not in mdl.spatial_dims}
                out = (full, rg.dims, coords)
            out[0][tuple(sel.get(d, slice(None)) for d in out[1])] = rg.values
        full, out_dims, coords = out
        import xarray as xr

        xc = {k: v for k, v in X.coords.items() if set(v.dims) <= set(out_dims)}
        xc.update({k: ((k,), v) for k, v in coords.items() if k in out_dims and k not in xc})
        res = xr.DataArray(full, dims=out_dims, coords=xc)
        try:  # like the reference's map_blocks result: same spatial chunk structure (needs dask; without it the eager array)
            return res.chunk({d: mdl.chunksizes[d] for d in mdl.spatial_dims})
        except Exception:  # noqa: BLE001
            return res

    def _batched(self):
        m = self._model
        if isinstance(m, BcsdBase):
            # every cell's copy runs _pre_fit in the reference ('daily_nasa-nex' swaps the grouper class in, bcsd.py:34-44);
            # the prototype stays untouched
            self._bcsd_proto = copy.deepcopy(m)
            self._bcsd_proto._pre_fit()
            check_supported(self._bcsd_proto)
            return "bcsd"
        if isinstance(m, (PureAnalog, AnalogRegression)):
            if isinstance(m, AnalogRegression):
                m._check()
            else:
                from .gard import check_tree_kwargs

                check_tree_kwargs(m)
            return "analog"
        if isinstance(m, PureRegression):
            m._check()
            return "linreg"
        if isinstance(m, QuantileMappingReressor):
            check_extrapolate(m.extrapolate)
            m._engine_code()
            return "qm"
        if isinstance(m, CunnaneTransformer):
            m._check()
            return "cunnane"
        if isinstance(m, QuantileMapper):
            m._check()
            return "qmapper"
        return None

    # ------------------------------------------------------------------------------------------
    def fit(self, X, *args, **kwargs):
        """Fit the model for every cell (core.py:225-264)."""
        kws = {"along_dim": self._dim, "feature_dim": DEFAULT_FEATURE_DIM} | kwargs
        if len(args) > 1:
            raise ValueError(f"Expected at most 1 positional argument, got {len(args)}")
        feature_dim = kws["feature_dim"]
        blocks = self._spatial_chunks(X, feature_dim)
        if blocks is not None:  # chunked input (dask-backed xarray, GridArray.chunk): block by block (core.py:256-262)
            dims, sizes, chunksizes = blocks
            fitted = []
            for sel in _block_slices(dims, chunksizes):
                child = PointWiseDownscaler(copy.deepcopy(self._model), self._dim)
                child.fit(_unchunked(_isel(X, sel)), *[_unchunked(_isel(a, sel)) for a in args], **kwargs)
                fitted.append((sel, child))
            self._models = _BlockedModels(dims, [sizes[d] for d in dims], chunksizes, fitted)
            return
        Xg, _ = self._to_feature_x(X, feature_dim)
        yg = None
        spatial_dims, spatial_shape = Xg.dims[2:], Xg.shape[2:]
        if args:
            # the reference selects y[index] by dimension *name* for every cell of X (core.py:86-93): align y to X's cell order
            yg, _ = _to_grid(args[0], feature_dim)
            if set(yg.dims) != {self._dim, *spatial_dims}:
                raise ValueError(f"y has dims {yg.dims}; expected {(self._dim,) + tuple(spatial_dims)} (the spatial dims of X, no feature dim)")
            yg = yg.transpose(self._dim, *spatial_dims)
            if tuple(yg.shape[1:]) != tuple(spatial_shape) or yg.shape[0] != Xg.shape[0]:
                raise ValueError(f"y has sizes {yg.sizes}, X has {Xg.sizes}")
        T, F = Xg.shape[:2]
        C = int(np.prod(spatial_shape, dtype=np.int64)) if spatial_shape else 1
        Xv = np.ascontiguousarray(Xg.values, dtype=np.float64).reshape(T, F, C)
        mask = ~np.isnan(Xv[0, 0, :])  # core.py:35-37
        index = _time_index(Xg, self._dim)
        kind = self._batched()
        coords = {k: v for k, v in Xg.coords.items() if k in spatial_dims}
        if kind is None:
            self._models = self._fit_loop(Xg, yg, Xv, mask, index, feature_dim, spatial_dims, spatial_shape, coords,
                                          {k: v for k, v in kws.items() if k not in ("along_dim", "feature_dim")})
            return
        m = self._model
        if kind in ("cunnane", "qmapper"):  # transformers: fit(X) only (a y, if given, is ignored like in the reference)
            if F != 1:
                if kind == "cunnane":
                    raise ValueError("CunnaneTransformer.fit() only supports a single feature")
                raise ValueError(f"Found array with {F} features (shape=({T}, {F})) while a maximum of 1 is required")
            gm = (CunnaneGridModel(m.extrapolate, m.n_endpoints) if kind == "cunnane" else QuantileMapperGridModel(detrend=m.detrend)).fit(Xv[:, 0, :])
            gm.status_ = gm.state.export(with_y=False)["status"] if kind == "cunnane" else gm.status_
            self._raise_for_status(gm.status_, Xv[:, 0, :], Xv[:, 0, :])
            self._models = _BatchedModels(kind, gm, mask, spatial_dims, spatial_shape, coords)
            return
        if yg is None:
            raise TypeError(f"{type(self._model).__name__}.fit() missing 1 required positional argument: 'y'")
        yv = np.ascontiguousarray(yg.values, dtype=np.float64).reshape(T, C)
        if kind == "bcsd":
            if F != 1:
                msg = "BCSD only supports up to 4 features, found {}" if m._kind == _lib.BCSD_TAS else "BCSD only supports 1 feature, found {}"
                raise ValueError(msg.format(F))
            gm = self._bcsd_proto._new_grid()
            if Xg.dtype == np.float32 and yg.dtype == np.float32:  # float32 grids cross PCIe as float32 (widened in HBM, exact)
                gm.fit(np.ascontiguousarray(Xg.values).reshape(T, C), np.ascontiguousarray(yg.values).reshape(T, C), index)
            else:
                gm.fit(Xv[:, 0, :], yv, index)
            self._raise_for_status(gm.status_, Xv[:, 0, :], yv)
        elif kind == "qm":
            if F != 1:
                raise ValueError(f"Found array with {F} features (shape=({T}, {F})) while a maximum of 1 is required")
            if T < 2 * m.n_endpoints + 1:
                raise ValueError(f"Found array with {T} sample(s) (shape=({T}, 1)) while a minimum of {2 * m.n_endpoints + 1} is required.")
            gm = QmGridModel(m._engine_code(), m.extrapolate, n_endpoints=m.n_endpoints)
            gm.fit(Xv[:, 0, :], yv)
            gm.status_ = gm.state.export()["status"]
            self._raise_for_status(gm.status_, Xv[:, 0, :], yv)
        else:
            gm = RegressionGridModel(thresh=m.thresh).fit(Xv, yv) if kind == "linreg" else AnalogGridModel(m.n_analogs).fit(Xv, yv)
            gm.status_ = np.where(mask, 0, _lib.CELL_MASKED).astype(np.int32)
            if kind == "linreg" and m.thresh is not None:
                e = gm.export()
                live_one_class = mask & (e["status"] == _lib.CELL_ONE_CLASS)
                if (mask & (e["thresh_dropped"] | (e["status"] == _lib.CELL_ONE_CLASS))).any():  # gard.py:426-437, per cell
                    import warnings

                    warnings.warn("Found only one class while attempting logistic regression. Mutating attribute thresh")
                if live_one_class.any():  # the linear model of such a cell gets an empty sample (gard.py:439)
                    from .gard import NO_SAMPLES_MESSAGE

                    raise ValueError(NO_SAMPLES_MESSAGE.format(F=F))
            bad = mask & ~(np.isfinite(Xv).all(axis=(0, 1)) & np.isfinite(yv).all(axis=0))
            if bad.any():
                c = int(np.flatnonzero(bad)[0])
                self._raise_for_status(np.where(bad, _lib.CELL_NONFINITE, 0), Xv[:, :, c:c + 1].reshape(T, -1), yv[:, c:c + 1], cell=c)
        self._models = _BatchedModels(kind, gm, mask, spatial_dims, spatial_shape, coords)

    @staticmethod
    def _raise_for_status(status, Xv, yv, cell=None):
        """Raise like the reference would for the first offending cell (base.py:18-20, bcsd.py:140-141)."""
        bad = np.flatnonzero((status == _lib.CELL_NONFINITE) | (status == _lib.CELL_BAD_CLIMO))
        if not len(bad):
            return
        c = int(bad[0])
        if status[c] == _lib.CELL_BAD_CLIMO:
            raise ValueError("Invalid value in target climatology")
        xc = Xv if cell is not None else Xv[:, c]
        yc = yv if cell is not None else yv[:, c]
        for name, a in (("X", xc), ("y", yc)):
            if np.isnan(a).any():
                raise ValueError(f"Input {name} contains NaN.")
            if not np.isfinite(a).all():
                raise ValueError(f"Input {name} contains infinity or a value too large for dtype('float64').")
        raise ValueError("Input contains NaN.")

    def _fit_loop(self, Xg, yg, Xv, mask, index, feature_dim, spatial_dims, spatial_shape, coords, fit_kwargs):
        """The reference's per-cell loop (core.py:69-97) for estimators the engine does not batch."""
        T, F, C = Xv.shape
        columns = list(Xg.coords.get(feature_dim, [f"feature{i}" for i in range(F)]))
        yv = None if yg is None else np.asarray(yg.values).reshape(T, C)
        models = np.full(C, None, dtype=object)
        for c in range(C):
            mod = copy.deepcopy(self._model)
            if not mask[c]:
                continue
            xdf = _da_to_df(Xv[:, :, c], index, columns)
            if yv is not None:
                ydf = _da_to_df(yv[:, c:c + 1], index, [f"{feature_dim}_0"])
                models[c] = mod.fit(xdf, ydf, **fit_kwargs)
            else:
                models[c] = mod.fit(xdf, **fit_kwargs)
        return _BatchedModels("loop", models, mask, spatial_dims, spatial_shape, coords)

    def _align_to_fitted(self, Xg, feature_dim):
        """Bring the spatial dims of a predict / transform input into the fitted order (the reference looks the fitted models
        up by dimension name, core.py:110-141): same dim names required, any order accepted."""
        fitted = tuple(self._models.spatial_dims)
        if set(Xg.dims[2:]) != set(fitted):
            raise ValueError(f"spatial dims {Xg.dims[2:]} do not match the fitted grid's {fitted}")
        Xg = Xg.transpose(self._dim, feature_dim, *fitted)
        if tuple(Xg.shape[2:]) != tuple(self._models.spatial_shape):
            raise ValueError(f"spatial shape {Xg.shape[2:]} does not match the fitted grid {self._models.spatial_shape}")
        return Xg

    # ------------------------------------------------------------------------------------------
    def predict(self, X, **kwargs):
        """Predict for every fitted cell (core.py:266-338); masked cells stay NaN."""
        if self._models is None:
            raise ValueError("PointWiseDownscaler is not fitted: call fit() first")
        if isinstance(self._models, _BlockedModels):
            return self._apply_blocks("predict", X, kwargs)
        kws = {"along_dim": self._dim, "feature_dim": DEFAULT_FEATURE_DIM} | kwargs
        feature_dim = kws["feature_dim"]
        Xg, was_x = self._to_feature_x(X, feature_dim)
        Xg = self._align_to_fitted(Xg, feature_dim)
        T, F = Xg.shape[:2]
        spatial_dims, spatial_shape = Xg.dims[2:], Xg.shape[2:]
        C = int(np.prod(spatial_shape, dtype=np.int64)) if spatial_shape else 1
        Xv = np.ascontiguousarray(Xg.values, dtype=np.float64).reshape(T, F, C)
        index = _time_index(Xg, self._dim)
        n_outputs = getattr(self._model, "n_outputs", 1)  # core.py:294-298
        output_names = getattr(self._model, "output_names", None)
        mdl = self._models
        coords = {k: v for k, v in Xg.coords.items() if k != feature_dim}
        if mdl.kind == "bcsd":
            if Xg.dtype == np.float32 and F == 1:  # in and out as float32, widened / narrowed on the device
                out, status = mdl.grid_model.predict(np.ascontiguousarray(Xg.values).reshape(T, C), index, out_dtype=np.float32)
            else:
                out, status = mdl.grid_model.predict(Xv[:, 0, :], index)
            self._raise_for_status(status, Xv[:, 0, :], Xv[:, 0, :])
            vals = out.reshape((T,) + tuple(spatial_shape)).astype(Xg.dtype, copy=False)
            res = GridArray(vals, (self._dim,) + spatial_dims, coords)
        elif mdl.kind == "qm":
            out, status = mdl.grid_model.predict(Xv[:, 0, :])
            self._raise_for_status(status, Xv[:, 0, :], Xv[:, 0, :])
            vals = out.reshape((T,) + tuple(spatial_shape)).astype(Xg.dtype, copy=False)
            res = GridArray(vals, (self._dim,) + spatial_dims, coords)
        elif mdl.kind in ("analog", "linreg"):
            m = self._model
            if mdl.kind == "linreg":
                out, status = mdl.grid_model.predict(Xv)
            elif isinstance(m, AnalogRegression):
                out, status = mdl.grid_model.predict_regression(Xv, m.thresh)
            else:
                out, status = mdl.grid_model.predict_pure(Xv, m.kind, m.thresh)
            if (status == _lib.CELL_NONFINITE).any():
                raise ValueError("Input X contains NaN.")
            if (status == _lib.CELL_ONE_CLASS).any():  # a query without any analog above the threshold (gard.py:204-207)
                from .gard import ONE_CLASS_MESSAGE

                raise ValueError(ONE_CLASS_MESSAGE)
            vals = out.reshape((T, 3) + tuple(spatial_shape)).astype(Xg.dtype, copy=False)
            coords[feature_dim] = np.array(output_names)
            res = GridArray(vals, (self._dim, feature_dim) + spatial_dims, coords)
        else:
            res = self._predict_loop(Xg, Xv, index, feature_dim, n_outputs, output_names, spatial_dims, spatial_shape,
                                     coords, {k: v for k, v in kws.items() if k not in ("along_dim", "feature_dim")})
        return _from_grid(res, was_x)

    def _predict_loop(self, Xg, Xv, index, feature_dim, n_outputs, output_names, spatial_dims, spatial_shape, coords, kw):
        """core.py:100-143 for estimators the engine does not batch."""
        T, F, C = Xv.shape
        columns = list(Xg.coords.get(feature_dim, [f"feature{i}" for i in range(F)]))
        shape = (T, C) if n_outputs == 1 else (T, n_outputs, C)
        y = np.full(shape, np.nan, dtype=Xg.dtype)
        for c in range(C):
            model = self._models.grid_model[c]
            if model is None:
                continue
            ydf = model.predict(_da_to_df(Xv[:, :, c], index, columns), **kw)
            y[..., c] = np.asarray(ydf).squeeze()
        if n_outputs == 1:
            return GridArray(y.reshape((T,) + tuple(spatial_shape)), (self._dim,) + spatial_dims, coords)
        coords[feature_dim] = np.array(output_names)
        return GridArray(y.reshape((T, n_outputs) + tuple(spatial_shape)), (self._dim, feature_dim) + spatial_dims, coords)

    # ------------------------------------------------------------------------------------------
    def transform(self, X, **kwargs):
        """Apply the fitted per-cell transformers (core.py:340-371)."""
        return self._transform(X, "transform", kwargs)

    def inverse_transform(self, X, **kwargs):
        """Apply the inverse of the fitted per-cell transformers (core.py:373-403)."""
        return self._transform(X, "inverse_transform", kwargs)

    def _transform(self, X, direction, kwargs):
        """core.py:146-171 (``_transform_wrapper``): same dims / shape as the feature-normalised X, masked cells NaN."""
        if self._models is None:
            raise ValueError("PointWiseDownscaler is not fitted: call fit() first")
        if isinstance(self._models, _BlockedModels):
            return self._apply_blocks(direction, X, kwargs)
        kind = self._models.kind
        if kind not in ("loop", "cunnane", "qmapper") or (kind == "qmapper" and direction != "transform"):
            raise AttributeError(f"{type(self._model).__name__} has no {direction}()")
        kws = {"feature_dim": DEFAULT_FEATURE_DIM} | kwargs
        feature_dim = kws.pop("feature_dim")
        Xg, was_x = self._to_feature_x(X, feature_dim)
        Xg = self._align_to_fitted(Xg, feature_dim)
        T, F = Xg.shape[:2]
        spatial_shape = Xg.shape[2:]
        C = int(np.prod(spatial_shape, dtype=np.int64)) if spatial_shape else 1
        if kind != "loop":  # one batched launch for the whole grid
            if F != 1:
                raise ValueError(f"{type(self._model).__name__}.{direction}() only supports a single feature")
            Xb = np.ascontiguousarray(Xg.values, dtype=np.float64).reshape(T, C)
            out, status = getattr(self._models.grid_model, direction)(Xb)
            self._raise_for_status(status, Xb, Xb)
            if kind == "cunnane" and direction == "transform" and np.isinf(out[:, self._models.mask]).any():
                raise AttributeError("'numpy.ndarray' object has no attribute 'values' (CunnaneTransformer.transform of values "
                                     f"outside the fitted range with extrapolate={self._model.extrapolate!r}: quantile.py:497)")
            out = out.reshape(Xg.shape).astype(Xg.dtype, copy=False)
            return _from_grid(GridArray(out, Xg.dims, dict(Xg.coords)), was_x)
        Xv = np.asarray(Xg.values).reshape(T, F, C)
        index = _time_index(Xg, self._dim)
        columns = list(Xg.coords.get(feature_dim, [f"feature{i}" for i in range(F)]))
        out = np.full((T, F, C), np.nan, dtype=Xg.dtype)
        for c in range(C):
            model = self._models.grid_model[c]
            if model is None:
                continue
            res = getattr(model, direction)(_da_to_df(Xv[:, :, c], index, columns), **kws)
            out[:, :, c] = np.asarray(res).reshape(T, F)
        return _from_grid(GridArray(out.reshape(Xg.shape), Xg.dims, dict(Xg.coords)), was_x)

    # ------------------------------------------------------------------------------------------
    def _cell_model(self, c, cache):
        """The fitted estimator of cell ``c`` of an engine-batched grid, rebuilt from the exported state: what the
        reference keeps per cell in its object array (core.py:81-96)."""
        mdl = self._models
        if mdl.kind == "loop":
            return mdl.grid_model[c]
        if not mdl.mask[c]:
            return None
        m = self._model
        if mdl.kind == "bcsd":
            e = cache.setdefault("e", mdl.grid_model.export())
            est = copy.deepcopy(self._bcsd_proto)
            est._adopt(e, c)
            est.n_features_in_ = 1
            return est
        if mdl.kind == "linreg":
            e = cache.setdefault("e", mdl.grid_model.export())
            est = copy.deepcopy(m)
            est.n_features_in_ = e["coef"].shape[0]
            if est.thresh is not None and e["thresh_dropped"][c]:
                est.thresh = None  # gard.py:437
            est._adopt(e, c)
            return est
        if mdl.kind == "analog":
            est = copy.deepcopy(m)
            est.k_ = mdl.grid_model.k_
            est.n_features_in_ = mdl.grid_model.state.info()["F"]
            return est
        if mdl.kind == "qm":
            e = cache.setdefault("e", mdl.grid_model.state.export())
            est = copy.deepcopy(m)
            est._X_cdf = est._extended(e["x_sorted"][c])
            est._y_cdf = est._extended(e["y_sorted"][c])
            est.n_features_in_ = 1
            return est
        if mdl.kind == "cunnane":
            from .quantile import Cdf, plotting_positions

            e = cache.setdefault("e", mdl.grid_model.state.export(with_y=False))
            est = copy.deepcopy(m)
            est.cdf_ = Cdf(plotting_positions(e["x_sorted"].shape[1]), e["x_sorted"][c])
            est.n_features_in_ = 1
            return est
        if mdl.kind == "qmapper":
            from .quantile import Cdf, FittedCunnane, plotting_positions

            e = cache.setdefault("e", mdl.grid_model.state.export())
            est = copy.deepcopy(m)
            vals = e["y_sorted"][c]
            est.x_cdf_fit_ = FittedCunnane(Cdf(plotting_positions(len(vals)), vals))
            if est.detrend:
                from .trend import FittedLine, FittedTrend

                est.x_trend_fit_ = FittedTrend(FittedLine(np.array([[e["y_trend"][c, 0, 0]]]), np.array([e["y_trend"][c, 0, 1]])))
            est.n_features_in_ = 1
            return est
        raise NotImplementedError(f"get_attr is not available for {mdl.kind} grids")

    def get_attr(self, key, dtype=np.float64, template_output=None):
        """Get attribute values specified in ``key`` from each of the pointwise models (core.py:405-425, 174-197): an array
        shaped like the model grid, or like ``template_output`` (whose non-spatial dims receive array-valued attributes).
        Engine-batched grids rebuild the per-cell fitted attributes from the exported state.  Extension: without a template,
        the BCSD climatologies ``y_climo_`` / ``_x_climo`` come back as [group, *spatial] fields."""
        mdl = self._models
        if mdl is None:
            raise ValueError("PointWiseDownscaler is not fitted: call fit() first")
        if isinstance(mdl, _BlockedModels):
            return self._get_attr_blocks(key, dtype, template_output)
        if mdl.kind == "bcsd" and key in ("y_climo_", "_x_climo") and template_output is None:
            e = mdl.grid_model.export()
            a = e["y_climo" if key == "y_climo_" else "x_climo"].T.astype(dtype)  # [G, C]
            a = np.where(mdl.mask[None, :], a, np.nan)
            coords = dict(mdl.coords)
            coords["group"] = e["keys"]
            return GridArray(a.reshape((a.shape[0],) + tuple(mdl.spatial_shape)), ("group",) + tuple(mdl.spatial_dims), coords)
        sp_dims, sp_shape = tuple(mdl.spatial_dims), tuple(mdl.spatial_shape)
        was_x = False
        if template_output is None:
            dims, shape, coords = sp_dims, sp_shape, dict(mdl.coords)
        else:
            if _is_xarray(template_output):
                import xarray as xr

                was_x = True
                if isinstance(template_output, xr.Dataset):  # core.py:184-186
                    template_output = template_output[list(template_output.data_vars)[0]]
            elif isinstance(template_output, GridDataset):
                template_output = template_output[list(template_output)[0]]
            tg, _ = _to_grid(template_output, DEFAULT_FEATURE_DIM)
            dims, shape, coords = tg.dims, tg.shape, dict(tg.coords)
            if not set(sp_dims) <= set(dims):
                raise ValueError(f"template_output dims {dims} do not contain the model grid's dims {sp_dims}")
        other = tuple(d for d in dims if d not in sp_dims)
        C = int(np.prod(sp_shape, dtype=np.int64)) if sp_shape else 1
        sizes = dict(zip(dims, shape))
        with np.errstate(invalid="ignore"):
            flat = np.full(tuple(sizes[d] for d in other) + (C,), np.nan, dtype=dtype)  # core.py:191
        cache = {}
        for c in range(C):
            est = self._cell_model(c, cache)
            if est is None:
                continue
            flat[..., c] = np.asarray(getattr(est, key)).reshape(flat.shape[:-1])  # core.py:194-196
        full = flat.reshape(tuple(sizes[d] for d in other) + sp_shape)
        g = GridArray(full, other + sp_dims, coords).transpose(*dims)
        return _from_grid(g, was_x)

    def _get_attr_blocks(self, key, dtype, template_output):
        mdl = self._models
        out = None
        for sel, child in mdl.blocks:
            tmpl = None if template_output is None else _unchunked(_isel(template_output, sel))
            res, _ = _to_grid(child.get_attr(key, dtype, tmpl), DEFAULT_FEATURE_DIM)
            if out is None:
                sizes = dict(res.sizes)
                sizes.update(mdl.sizes)
                with np.errstate(invalid="ignore"):
                    full = np.full([sizes[d] for d in res.dims], np.nan, dtype=dtype)
                out = (full, res.dims, {k: v for k, v in res.coords.items() if k not in mdl.spatial_dims})
            out[0][tuple(sel.get(d, slice(None)) for d in out[1])] = res.values
        return _from_grid(GridArray(*out), template_output is not None and _is_xarray(template_output))

    def __repr__(self):
        summary = [f"<skdownscale.{self.__class__.__name__}>", f"  Fit Status: {self._models is not None}",
                   f"  Model:\n    {self._model}"]
        return "\n".join(summary)
