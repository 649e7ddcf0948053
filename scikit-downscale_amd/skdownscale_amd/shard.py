"""Cell-axis sharding across the GPUs of one node.

Cells are independent (each gets its own estimator in the reference, core.py:87), so the grid is
block-partitioned over ranks with no exchange during fit/predict; the only communication is the
gather of the predicted field ``out[Tp, C_local]`` to the root.  ``torch.distributed`` is launcher
plumbing here (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in CPU tests); the engine
itself never imports torch.
"""
from __future__ import annotations

import numpy as np


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


def gather_field(local, n_cells: int, dst: int = 0, group=None):
    """Gather ``local`` [..., C_local] (torch tensor, cells on the last axis) to ``dst``.

    Ragged blocks are padded to the widest block for the collective and trimmed on the root.
    Returns the [..., n_cells] tensor on ``dst`` and None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = cell_partition(n_cells, world)
    width = max(e - s for s, e in bounds)
    lead = tuple(local.shape[:-1])
    send = local
    if local.shape[-1] != width:
        send = torch.zeros(lead + (width,), dtype=local.dtype, device=local.device)
        send[..., : local.shape[-1]] = local
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[..., : e - s] for b, (s, e) in zip(bufs, bounds)], dim=-1)


def assemble(blocks, n_cells: int):
    """NumPy twin of the root-side concatenation (used by tests)."""
    out = np.concatenate(blocks, axis=-1)
    assert out.shape[-1] == n_cells
    return out
