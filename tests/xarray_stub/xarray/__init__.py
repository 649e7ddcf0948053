"""Minimal stand-in for xarray -- TEST INFRASTRUCTURE ONLY.

xarray is not installed in the build container nor on the GPU box, so the xarray branches of
``skdownscale_amd.core`` (``_to_grid`` / ``_from_grid`` / block selection of chunked inputs) would never execute.  This module
implements just the part of the ``xarray.DataArray`` / ``xarray.Dataset`` surface those branches use, with xarray's semantics
(dims + coords, ``isel``, ``to_array``, chunk metadata as dask-backed objects report it).  ``tests/conftest.py`` puts it on
``sys.path`` only when the real package cannot be imported; where xarray is installed the same tests run against it.
"""
from __future__ import annotations

import numpy as np

__version__ = "0.0-stub"


class _Coord:
    """a coordinate variable: values along the dimension of the same name"""

    def __init__(self, values, name=None):
        self.values = np.asarray(values)
        self.ndim = self.values.ndim
        self.dims = (name,) if self.ndim == 1 else ()


def _block_lengths(size, n):
    if isinstance(n, (tuple, list)):  # explicit block lengths
        assert sum(n) == size
        return tuple(int(b) for b in n)
    n = size if n in (-1, None) else int(n)
    out, left = [], size
    while left > 0:
        out.append(min(n, left))
        left -= n
    return tuple(out)


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None, _chunks=None):
        self.values = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self.values.ndim))
        assert len(self.dims) == self.values.ndim
        self.coords = {}
        for k, v in (coords or {}).items():
            if isinstance(v, tuple) and len(v) == 2 and not np.isscalar(v[0]) and isinstance(v[0], (tuple, list, str)):
                v = v[1]  # (dims, values) form
            self.coords[k] = v if isinstance(v, _Coord) else _Coord(v, k)
        self.name = name
        self._chunks = _chunks  # dim -> block lengths, like a dask-backed array

    @property
    def sizes(self):
        return dict(zip(self.dims, self.values.shape))

    @property
    def shape(self):
        return self.values.shape

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def chunks(self):
        if self._chunks is None:
            return None
        return tuple(self._chunks.get(d, (self.sizes[d],)) for d in self.dims)

    @property
    def chunksizes(self):
        return {} if self._chunks is None else {d: self._chunks.get(d, (self.sizes[d],)) for d in self.dims}

    def chunk(self, chunks):
        return DataArray(self.values, self.dims, self.coords, self.name,
                         {d: _block_lengths(self.sizes[d], chunks.get(d, -1)) for d in self.dims})

    def compute(self):
        return DataArray(self.values, self.dims, self.coords, self.name)

    def isel(self, **sel):
        index = tuple(sel.get(d, slice(None)) for d in self.dims)
        coords = {}
        for k, c in self.coords.items():
            coords[k] = _Coord(c.values[sel[k]], k) if (k in sel and k in self.dims and c.ndim == 1) else c
        chunks = None
        if self._chunks is not None:  # a block of a chunked array: the selected dims become single blocks
            vals = self.values[index]
            chunks = {d: ((vals.shape[i],) if d in sel else self._chunks.get(d, (vals.shape[i],))) for i, d in enumerate(self.dims)}
        return DataArray(self.values[index], self.dims, coords, self.name, chunks)

    def transpose(self, *dims):
        order = [self.dims.index(d) for d in dims]
        return DataArray(self.values.transpose(order), dims, self.coords, self.name, self._chunks)


class Dataset:
    def __init__(self, data_vars=None, coords=None):
        self.data_vars = {}
        for k, v in (data_vars or {}).items():
            if isinstance(v, DataArray):
                self.data_vars[k] = v
            else:
                dims, values = v
                own = {c: cv for c, cv in (coords or {}).items() if c in dims}
                self.data_vars[k] = DataArray(values, dims, own, k)
        self.coords = {k: _Coord(v, k) for k, v in (coords or {}).items()}

    def __iter__(self):
        return iter(self.data_vars)

    def __getitem__(self, key):
        return self.data_vars[key]

    def _first(self):
        return next(iter(self.data_vars.values()))

    @property
    def dims(self):
        return self._first().dims

    @property
    def sizes(self):
        return self._first().sizes

    @property
    def chunks(self):
        return self._first().chunksizes or None

    @property
    def chunksizes(self):
        return self._first().chunksizes

    def chunk(self, chunks):
        return Dataset({k: v.chunk(chunks) for k, v in self.data_vars.items()})

    def compute(self):
        return Dataset({k: v.compute() for k, v in self.data_vars.items()})

    def isel(self, **sel):
        return Dataset({k: v.isel(**sel) for k, v in self.data_vars.items()})

    def to_array(self, dim="variable"):
        first = self._first()
        vals = np.stack([v.values for v in self.data_vars.values()], axis=0)
        coords = dict(first.coords)
        coords[dim] = _Coord(np.array(list(self.data_vars)), dim)
        chunks = None if first._chunks is None else {**first._chunks, dim: (len(self.data_vars),)}
        return DataArray(vals, (dim,) + first.dims, coords, None, chunks)
