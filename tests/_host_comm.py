"""Test transport: the product's gather layout (skdownscale_amd.shard._GatherLayout) over the rendezvous sockets, so that
``gather_field`` and ``ShardedPointWiseDownscaler`` run with several processes on a machine without GPUs.  Not part of the
package: the product gathers over RCCL (shard.Communicator)."""
import numpy as np

from skdownscale_amd.shard import _GatherLayout


def _bytes_of(arr):
    a = np.ascontiguousarray(arr).reshape(-1)
    return memoryview(a).cast("B") if a.size else memoryview(b"")


def gather_bytes(rdv, payload, into):
    """every rank's ``payload`` (ndarray) to rank 0 into the writable memoryviews ``into`` (sizes known to the root)"""
    st = rdv._struct
    view = _bytes_of(payload)
    if rdv.rank != 0:
        rdv.root.sendall(st.pack("<q", view.nbytes))
        if view.nbytes:
            rdv.root.sendall(view)
        return
    if view.nbytes:
        into[0][:] = view
    for r, conn in rdv.peers.items():
        (n,) = st.unpack("<q", rdv._recv(conn, 8))
        buf = into[r]
        if n != buf.nbytes:  # (checked before anything is received or allocated)
            raise ValueError(f"rank {r} sent {n} bytes, expected {buf.nbytes}")
        got = 0
        while got < n:
            k = conn.recv_into(buf[got:], n - got)
            if k == 0:
                raise ConnectionError("rendezvous peer closed the connection")
            got += k


class HostCommunicator(_GatherLayout):
    def __init__(self, rendezvous):
        self.rdv = rendezvous
        self.rank, self.world = rendezvous.rank, rendezvous.world

    def barrier(self):
        self.rdv.barrier()

    def allreduce_max(self, value):
        return self.rdv.allreduce_max(value)

    def _alloc(self, n):
        return np.empty(int(n), dtype=np.float64)

    @staticmethod
    def _nbytes(buf):
        return buf.nbytes

    @staticmethod
    def _view(buf, off, shape):
        n = int(np.prod(shape, dtype=np.int64))
        return buf.reshape(-1)[int(off):int(off) + n].reshape(shape)  # NumPy views keep their base alive

    def _transport(self, local, T, cells, root_buffer, root, wait):
        if root != 0:
            raise ValueError("the socket transport gathers to rank 0 (the root of the rendezvous star)")
        local = np.ascontiguousarray(local, dtype=np.float64)
        if self.world == 1:
            root_buffer.reshape(-1)[:local.size] = local.reshape(-1)
            return
        into = None
        if self.rank == 0:
            flat = root_buffer.reshape(-1).view(np.uint8)
            into, off = [], 0
            for r in range(self.world):
                n = T * int(cells[r]) * 8
                into.append(memoryview(flat[off:off + n]) if n else memoryview(bytearray(0)))
                off += n
        gather_bytes(self.rdv, local, into)

    def wait(self):
        pass

    def close(self):
        pass
