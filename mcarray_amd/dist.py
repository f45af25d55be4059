"""Sharding of independent arrays over the GPUs of one node (one process per GPU).

Arrays / streams never interact (no state is shared between module objects), so the path shards
embarrassingly: rank r owns a contiguous block of arrays, runs the whole path on them with no
collective inside the compute, and the only exchange is a gather of the per-frame DOA buffers
(8 bytes per frame per array) -- `torch.distributed` all_gather, RCCL over xGMI with backend
"nccl", gloo on CPU in the tests.
"""
import torch
import torch.distributed as dist


def partition(n_units, world):
    """Contiguous block sizes, first n_units % world ranks get one more."""
    base, rem = divmod(n_units, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def local_range(n_units, rank, world):
    sizes = partition(n_units, world)
    start = sum(sizes[:rank])
    return range(start, start + sizes[rank])


def array_seed(base_seed, global_array_index):
    """Synthetic-input seed of an array depends on its GLOBAL index only, so the data a given array
    sees does not depend on how many ranks the job has."""
    return int(base_seed) + int(global_array_index)


def gather_arrays(local, n_units, group=None):
    """local: [A_local, ...] result block of this rank -> [n_units, ...] on every rank, in array order.
    Handles ragged blocks by padding to the largest block."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = partition(n_units, world)
    amax = max(sizes)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block has %d arrays, partition says %d" % (local.shape[0], sizes[dist.get_rank(group)]))
    if local.shape[0] < amax:
        pad = torch.zeros((amax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
