"""Cell-axis sharding across the GPUs of one node.

Cells are independent (each gets its own estimator in the reference, core.py:87), so the grid is
block-partitioned over ranks with no exchange during fit/predict; the only communication is the
gather of the predicted field ``out[Tp, C_local]`` to the root.

``Communicator`` is the product path: one process per GPU, RCCL over xGMI through the engine's own C ABI
(``sd_comm_*``, csrc/sd_comm.hip) -- no PyTorch.  The ranks find each other through the launcher's environment
(``RANK``, ``WORLD_SIZE``, ``MASTER_ADDR``, ``MASTER_PORT``: what ``python -m torch.distributed.run`` or any other
launcher exports); rank 0 hands RCCL's 128-byte unique id to the others over a TCP socket.
``ShardedPointWiseDownscaler`` is the drop-in surface on top of it: the ranks' results stay on their GPUs, travel to the
root's GPU with one grouped ncclSend / ncclRecv, and cross PCIe once, on the root.  The layout logic (``_GatherLayout``) is
transport independent; tests/_host_comm.py plugs a socket transport under it so that it runs with several processes on a
machine without GPUs (tests/test_host.py) -- the package itself carries no host data plane.
"""
from __future__ import annotations

import ctypes
import os
import socket
import time

import numpy as np

ID_BYTES = 128
PORT_OFFSET = 17  # the id is served on MASTER_PORT + PORT_OFFSET (MASTER_PORT itself may belong to the launcher's store)


def exchange_unique_id(rank, world, make_id, addr=None, port=None, timeout=300.0):
    """Rank 0 calls ``make_id()`` (-> 128 bytes) and serves the result to the other ranks; every rank returns it."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + PORT_OFFSET)
    if world == 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        assert len(uid) == ID_BYTES
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        try:
            for _ in range(world - 1):
                conn, _peer = srv.accept()
                with conn:
                    conn.sendall(uid)
        finally:
            srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                buf = b""
                while len(buf) < ID_BYTES:
                    chunk = conn.recv(ID_BYTES - len(buf))
                    if not chunk:
                        break
                    buf += chunk
            if len(buf) == ID_BYTES:
                return buf
        except OSError:
            pass
        if time.time() > deadline:
            raise TimeoutError(f"rank {rank}: no unique id from {addr}:{port} within {timeout} s")
        time.sleep(0.05)


class Rendezvous:
    """Control plane of a one-node job: a TCP star around rank 0 (persistent connections), found through the launcher's
    ``MASTER_ADDR`` / ``MASTER_PORT`` (+ PORT_OFFSET).  Carries what is not worth a GPU collective -- RCCL's unique id, the
    barrier and the max-over-ranks of a wall-clock time -- and keeps a job's timing independent of the data-path library."""

    def __init__(self, rank, world, addr=None, port=None, timeout=300.0):
        import struct

        self._struct = struct
        self.rank, self.world = int(rank), int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + PORT_OFFSET)
        self.peers, self.root = {}, None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            try:
                while len(self.peers) < self.world - 1:
                    conn, _peer = srv.accept()
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    (r,) = struct.unpack("<i", self._recv(conn, 4))
                    if r < 1 or r >= self.world or r in self.peers:  # not one of this job's ranks (or a second claim of one)
                        conn.close()
                        continue
                    self.peers[r] = conn
            finally:
                srv.close()
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    conn = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise TimeoutError(f"rank {self.rank}: no rendezvous at {addr}:{port} within {timeout} s") from None
                    time.sleep(0.05)
            conn.settimeout(timeout)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.sendall(struct.pack("<i", self.rank))
            self.root = conn

    @classmethod
    def from_env(cls, timeout=300.0):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), timeout=timeout)

    @staticmethod
    def _recv(conn, n):
        buf = b""
        while len(buf) < n:
            chunk = conn.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += chunk
        return buf

    def broadcast(self, payload=None):
        """bytes from rank 0 to everybody (``payload`` is read on rank 0 only)"""
        if self.world == 1:
            return payload
        st = self._struct
        if self.rank == 0:
            msg = st.pack("<q", len(payload)) + payload
            for conn in self.peers.values():
                conn.sendall(msg)
            return payload
        (n,) = st.unpack("<q", self._recv(self.root, 8))
        return self._recv(self.root, n)

    def allreduce_max(self, value):
        """max of a float over the ranks (also a barrier: nobody returns before everybody has arrived)"""
        if self.world == 1:
            return float(value)
        st = self._struct
        if self.rank == 0:
            m = float(value)
            for conn in self.peers.values():
                m = max(m, st.unpack("<d", self._recv(conn, 8))[0])
            for conn in self.peers.values():
                conn.sendall(st.pack("<d", m))
            return m
        self.root.sendall(st.pack("<d", float(value)))
        return st.unpack("<d", self._recv(self.root, 8))[0]

    def barrier(self):
        self.allreduce_max(0.0)

    def close(self):
        for conn in list(self.peers.values()) + ([self.root] if self.root is not None else []):
            try:
                conn.close()
            except OSError:
                pass
        self.peers, self.root = {}, None


class _GatherLayout:
    """The gather of a predicted field to a root, shared by the RCCL and the socket transport.  Every rank holds a contiguous
    [T, C_r] block; the root receives all of them into ONE buffer laid out [rank][T][C_r] (ragged shards back to back, no
    concatenation copy) and gets one [T, C_r] view per rank.  Subclasses provide ``_alloc(n)`` (buffer of n float64),
    ``_view(buffer, offset, shape)`` (a view that keeps ``buffer`` alive), ``_nbytes(buffer)`` and ``_transport(...)``."""

    rank = 0
    world = 1

    def gather_field(self, local, cells, root=0, root_buffer=None, wait=True):
        """``cells`` = C_r of every rank.  On the root returns the list of per-rank views into ``root_buffer`` (allocated
        when None: sum(cells) * T doubles); elsewhere None."""
        cells = np.ascontiguousarray(cells, dtype=np.int64)
        if cells.shape != (self.world,):
            raise ValueError(f"cells: expected {self.world} entries (one per rank), got shape {cells.shape}")
        if not 0 <= int(root) < self.world:
            raise ValueError(f"root={root} outside [0, {self.world})")
        T = int(local.shape[0])
        ld = getattr(local, "ld", local.shape[-1])
        if len(local.shape) != 2 or ld != local.shape[1] or local.shape[1] != cells[self.rank]:
            raise ValueError("gather_field needs a contiguous [T, cells[rank]] field")
        views = None
        if self.rank == root:
            need = int(cells.sum()) * T
            if root_buffer is None:
                root_buffer = self._alloc(need)
            elif self._nbytes(root_buffer) < need * 8:
                raise ValueError(f"root_buffer holds {self._nbytes(root_buffer)} bytes, the gathered field needs {need * 8}")
            views, off = [], 0
            for r in range(self.world):
                views.append(self._view(root_buffer, off, (T, int(cells[r]))))  # (each view pins root_buffer)
                off += T * int(cells[r])
        self._transport(local, T, cells, root_buffer, int(root), wait)
        return views


class Communicator(_GatherLayout):
    """RCCL communicator of the engine (one rank per process / GPU)."""

    def __init__(self, ctx, rank, world, unique_id):
        from ._lib import check

        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        h = ctypes.c_void_p()
        check(ctx.lib.sd_comm_create(ctx.handle, ctypes.c_char_p(unique_id), self.rank, self.world, ctypes.byref(h)))
        self.handle = h

    @classmethod
    def from_env(cls, ctx, timeout=300.0, rendezvous=None):
        """rank / world / rendezvous address from the launcher's environment; with a ``Rendezvous`` the unique id travels over
        its connections (otherwise rank 0 serves it once on MASTER_PORT + PORT_OFFSET)"""
        from ._lib import check

        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))

        def make_id():
            buf = ctypes.create_string_buffer(ID_BYTES)
            check(ctx.lib.sd_comm_unique_id(buf))
            return buf.raw

        if rendezvous is not None:
            if rank == 0:
                try:
                    uid = make_id()
                except Exception:
                    rendezvous.broadcast(b"")  # (the other ranks are waiting for it: let them fail too instead of hanging)
                    raise
                rendezvous.broadcast(uid)
            else:
                uid = rendezvous.broadcast(None)
                if len(uid) != ID_BYTES:
                    from ._lib import EngineError

                    raise EngineError("rank 0 could not create the RCCL unique id")
            return cls(ctx, rank, world, uid)
        return cls(ctx, rank, world, exchange_unique_id(rank, world, make_id, timeout=timeout))

    def close(self):
        if self.handle is not None:
            self.ctx.lib.sd_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def rccl_info(self):
        """what RCCL itself reports: ``{"version", "ranks", "device"}`` (ncclGetVersion / ncclCommCount / ncclCommCuDevice)"""
        from ._lib import check

        v, n, d = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
        check(self.ctx.lib.sd_comm_rccl_info(self.handle, ctypes.byref(v), ctypes.byref(n), ctypes.byref(d)))
        return {"version": v.value, "ranks": n.value, "device": d.value}

    def barrier(self):
        from ._lib import check

        check(self.ctx.lib.sd_comm_barrier(self.handle))

    def allreduce_max(self, value):
        from ._lib import check

        out = ctypes.c_double()
        check(self.ctx.lib.sd_comm_allreduce_max(self.handle, float(value), ctypes.byref(out)))
        return out.value

    # ---- transport: grouped ncclSend / ncclRecv into the root's buffer (csrc/sd_comm.hip) ----
    def _alloc(self, n):
        return self.ctx.empty((int(n),))

    @staticmethod
    def _nbytes(buf):
        return buf.nbytes

    def _view(self, buf, off, shape):
        from .engine import DeviceArray

        return DeviceArray(self.ctx, shape, np.float64, dptr=buf.ptr + int(off) * 8, owner=False, base=buf)

    def _transport(self, local, T, cells, root_buffer, root, wait):
        from ._lib import check

        check(self.ctx.lib.sd_comm_gather_field(self.handle, local.vptr, T, cells.ctypes.data_as(ctypes.c_void_p),
                                                None if root_buffer is None else root_buffer.vptr, root, 1 if wait else 0))

    def wait(self):
        from ._lib import check

        check(self.ctx.lib.sd_comm_wait(self.handle))


def cell_partition(n_cells: int, world: int):
    """Contiguous blocks, sizes differ by at most one: list of (start, stop)."""
    base, rem = divmod(int(n_cells), int(world))
    bounds, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        bounds.append((s, e))
        s = e
    return bounds


def local_cells(n_cells: int, world: int, rank: int):
    return cell_partition(n_cells, world)[rank]


def _cell_block(values, n_lead, s, e):
    """Cells [s, e) of the flattened spatial axes of ``values`` ([*lead, *spatial], any strides) as a contiguous [*lead, e - s]
    array.  Only the slab of the first spatial axis that holds the block is ever made contiguous: flattening a transposed
    (or memory-mapped) field first would copy the whole grid on every rank (117 GB per field at 1 M cells x 14 600 steps)."""
    lead_shape, sp_shape = values.shape[:n_lead], values.shape[n_lead:]
    if e <= s:
        return np.empty(lead_shape + (0,), dtype=values.dtype)
    if not sp_shape:
        return np.ascontiguousarray(values.reshape(lead_shape + (1,))[..., s:e])
    inner = int(np.prod(sp_shape[1:], dtype=np.int64))  # cells per step of the first spatial axis
    i0, i1 = s // inner, (e - 1) // inner + 1
    slab = values[(slice(None),) * n_lead + (slice(i0, i1),)]  # a view
    slab = np.ascontiguousarray(slab).reshape(lead_shape + ((i1 - i0) * inner,))
    return np.ascontiguousarray(slab[..., s - i0 * inner:e - i0 * inner])


class ShardedPointWiseDownscaler:
    """``PointWiseDownscaler`` over the GPUs of one node: every rank (one process per GPU, same script, same inputs) fits and
    predicts a contiguous block of the grid's cells (``cell_partition`` of the flattened spatial dims, in the spatial dim order
    of the fitted ``X``) on its own engine, and the predicted field is gathered to rank 0 with ``comm.gather_field``.
    Replaces the reference's ``map_blocks`` over dask workers (core.py:256-262, 300-336) for one node.

    With a ``Communicator`` (RCCL) the gather runs GPU to GPU: BCSD predictions never leave the device before it (the
    shard of ``X`` is uploaded, predicted and handed to the gather as a resident [T, C_r] field), results of the other
    estimators are uploaded for it; rank 0 then downloads the gathered [rank][T][C_r] buffer once.

    ``fit`` / ``predict`` / ``transform`` take what ``PointWiseDownscaler`` takes (GridArray, ndarray, xarray) and align their
    arguments by dimension NAME like it does (core.py:86-93, 110-141): ``y`` at fit and ``X`` at predict may order their
    spatial dims differently from the fitted ``X``.  ``predict`` returns the full result on rank 0 (dtype of the input, as
    ``PointWiseDownscaler``) and None on the other ranks."""

    def __init__(self, model, dim="time", comm=None):
        from .core import PointWiseDownscaler

        self.comm = comm
        self.rank = 0 if comm is None else comm.rank
        self.world = 1 if comm is None else comm.world
        self._dim = dim
        self._inner = PointWiseDownscaler(model, dim)
        self._layout = None  # (spatial dims, spatial shape, cells, spatial coords) of the fitted X

    # -- the cells of this rank as a [time, (feature), cell] grid ---------------------------------------------------------
    def _shard(self, X, feature_dim, layout=None):
        """``layout`` None: X defines the spatial dim order; else X is brought into the fitted order by dim name and must
        have the fitted spatial shape."""
        from .core import DEFAULT_FEATURE_DIM, GridArray, _to_grid

        fd = feature_dim or DEFAULT_FEATURE_DIM
        g, _ = _to_grid(X, fd)
        lead = [d for d in g.dims if d in (self._dim, fd)]
        spatial = [d for d in g.dims if d not in lead]
        if layout is not None:
            want_dims, want_shape = layout[0], layout[1]
            if set(spatial) != set(want_dims):
                raise ValueError(f"spatial dims {tuple(spatial)} do not match the fitted grid's {tuple(want_dims)}")
            spatial = list(want_dims)
        g = g.transpose(*lead, *spatial)
        sp_shape = g.shape[len(lead):]
        if layout is not None and tuple(sp_shape) != tuple(layout[1]):
            raise ValueError(f"spatial shape {tuple(sp_shape)} does not match the fitted grid {tuple(layout[1])}")
        C = int(np.prod(sp_shape, dtype=np.int64)) if sp_shape else 1
        s, e = cell_partition(C, self.world)[self.rank]
        coords = {k: v for k, v in g.coords.items() if k in lead}
        local = GridArray(_cell_block(np.asarray(g.values), len(lead), s, e), tuple(lead) + ("cell",), coords)
        return local, (tuple(spatial), tuple(sp_shape), C, {k: v for k, v in g.coords.items() if k in spatial})

    def fit(self, X, *args, **kwargs):
        fd = kwargs.get("feature_dim")
        Xl, self._layout = self._shard(X, fd)
        self._inner.fit(Xl, *[self._shard(a, fd, self._layout)[0] for a in args], **kwargs)
        return self

    def _resident_bcsd(self, Xl, kwargs):
        """BcsdTemperature / BcsdPrecipitation predict of this rank's shard with the result left on the GPU: (DeviceArray
        [T, C_r], lead dims, lead shape, lead coords, dtype) or None when the fitted model is not a batched BCSD grid."""
        from .core import DEFAULT_FEATURE_DIM, _time_index

        mdl = getattr(self._inner, "_models", None)
        if getattr(mdl, "kind", None) != "bcsd" or not hasattr(self.comm, "ctx"):
            return None
        fd = kwargs.get("feature_dim") or DEFAULT_FEATURE_DIM
        Xg, _ = self._inner._to_feature_x(Xl, fd)
        Xg = self._inner._align_to_fitted(Xg, fd)
        T, F = Xg.shape[:2]
        Cl = int(np.prod(Xg.shape[2:], dtype=np.int64))
        if F != 1 or Cl == 0:
            return None
        Xv = np.ascontiguousarray(Xg.values, dtype=np.float64).reshape(T, Cl)
        index = _time_index(Xg, self._dim)
        out, status = mdl.grid_model.predict(self.comm.ctx.to_device(Xv), index)  # DeviceArray in -> DeviceArray out
        self._inner._raise_for_status(status, Xv, Xv)
        coords = {k: v for k, v in Xg.coords.items() if k == self._dim}
        return out, (self._dim,), (T,), coords, Xg.dtype

    def _run(self, method, X, **kwargs):
        from .core import GridArray, _to_grid

        if self._layout is None:
            raise ValueError("ShardedPointWiseDownscaler is not fitted: call fit() first")
        fd = kwargs.get("feature_dim")
        Xl, _ = self._shard(X, fd, self._layout)
        spatial, sp_shape, C, sp_coords = self._layout
        cells = np.array([e - s for s, e in cell_partition(C, self.world)], dtype=np.int64)
        on_gpu = self.comm is not None and hasattr(self.comm, "ctx")
        resident = self._resident_bcsd(Xl, kwargs) if (method == "predict" and on_gpu) else None
        if resident is not None:
            local, lead_dims, lead_shape, lead_coords, dtype = resident
        else:
            res, _ = _to_grid(getattr(self._inner, method)(Xl, **kwargs), fd or "variable")
            res = res.transpose(*[d for d in res.dims if d != "cell"], "cell")
            lead_dims, lead_shape, dtype = res.dims[:-1], res.shape[:-1], res.dtype
            lead_coords = {k: v for k, v in res.coords.items() if k in lead_dims}
            rows = int(np.prod(lead_shape, dtype=np.int64))
            local = np.ascontiguousarray(np.asarray(res.values, dtype=np.float64).reshape(rows, res.shape[-1]))
        if self.comm is None or (self.world == 1 and not on_gpu):
            full = local
        else:  # (a one-rank RCCL communicator takes the same path: the gather is a device copy into the root buffer)
            if on_gpu and resident is None:
                local = self.comm.ctx.to_device(local)  # (results of the other estimators are host arrays: up for the gather)
            views = self.comm.gather_field(local, cells, 0)
            if self.rank != 0:
                return None
            if on_gpu:
                # one download of the root's [rank][T][C_r] buffer; the per-rank blocks are views of the host copy
                rows = int(views[0].shape[0])
                root = getattr(views[0], "base", None)  # the root buffer [rank][T][C_r] every per-rank view was cut from
                host = root.to_host().reshape(-1) if root is not None and root.nbytes == 8 * rows * int(cells.sum()) else None
                if host is None:
                    views = [v.to_host() for v in views]
                else:
                    blocks, off = [], 0
                    for c in cells:
                        blocks.append(host[off:off + rows * int(c)].reshape(rows, int(c)))
                        off += rows * int(c)
                    views = blocks
            full = np.concatenate(views, axis=1)  # per-rank blocks side by side = the flattened cell axis
        coords = dict(lead_coords)
        coords.update(sp_coords)
        full = np.asarray(full).reshape(tuple(lead_shape) + tuple(sp_shape)).astype(dtype, copy=False)
        return GridArray(full, tuple(lead_dims) + tuple(spatial), coords)

    def predict(self, X, **kwargs):
        return self._run("predict", X, **kwargs)

    def transform(self, X, **kwargs):
        return self._run("transform", X, **kwargs)

    def inverse_transform(self, X, **kwargs):
        return self._run("inverse_transform", X, **kwargs)
