"""TEST SCAFFOLDING (not part of the product): pad-and-trim gather of a cell-sharded torch tensor with torch.distributed
(gloo on CPU).  The product's gather is skdownscale_amd.shard.Communicator / HostCommunicator."""
import numpy as np
import torch
import torch.distributed as dist

from skdownscale_amd.shard import cell_partition


def gather_field(local, n_cells: int, dst: int = 0, group=None):
    """Gather ``local`` [..., C_local] (cells on the last axis) to ``dst``: ragged blocks are padded to the widest block for
    the collective and trimmed on the root.  Returns the [..., n_cells] tensor on ``dst`` and None elsewhere."""
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
    out = np.concatenate(blocks, axis=-1)
    assert out.shape[-1] == n_cells
    return out
