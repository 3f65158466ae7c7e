"""Sharding of independent units (frames, frame pairs) over ranks: contiguous blocks, results concatenated in
unit order, no collective on the per-frame path.  The one exchange step of the path - replicating a key-frame
descriptor set for many-to-many matching - is an all-gather of uint8 rows."""


def shard_range(n_units, rank, world):
    """Units [lo, hi) owned by `rank`: block i*n/world .. (i+1)*n/world."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (rank * n_units) // world, ((rank + 1) * n_units) // world


def all_gather_descriptors(local_desc, dist):
    """Replicates per-rank descriptor blocks ([k_local, n, 32] uint8 torch tensors of equal shape) on every rank
    (RCCL all-gather on GPUs, gloo on CPU), straight into one [world * k_local, n, 32] tensor."""
    import torch
    world = dist.get_world_size()
    local = local_desc.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out
