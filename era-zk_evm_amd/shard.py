"""Multi-GPU sharding of a batch.

VM instances share no state (every reference `VmState` owns its oracles by value, mod.rs:167-174),
so ranks own contiguous blocks of instances and run them with NO data-path collective.  The only
exchange is the final one: an all-gather of the per-instance queue digests (3 queues x 4 x u64 per
instance) and an all-reduce of the run counters — one collective each, default algorithm: at
<= 400 KB per rank, xGMI link bandwidth is irrelevant (RCCL when the backend is "nccl", gloo on CPU).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """contiguous block [first, first+count) of rank `rank`; blocks differ by at most one instance"""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def final_reduce(local_digests, local_counters, group=None):
    """local_digests: int64 tensor [n_local, 3, 4] (bit pattern of the u64 field elements);
    local_counters: int64 tensor [k] (cycles, mem, log, aux, ...).
    Returns (all_digests [sum n_local, 3, 4] in rank order, summed counters).  Ranks may own different
    numbers of instances (ragged): sizes are exchanged first."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_digests.clone(), local_counters.clone()
    n_local = torch.tensor([local_digests.shape[0]], dtype=torch.int64, device=local_digests.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    pad = torch.zeros((n_max,) + tuple(local_digests.shape[1:]), dtype=local_digests.dtype, device=local_digests.device)
    pad[: local_digests.shape[0]] = local_digests
    gathered = torch.empty((world * n_max,) + tuple(local_digests.shape[1:]), dtype=local_digests.dtype, device=local_digests.device)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    parts = [gathered[r * n_max: r * n_max + sizes[r]] for r in range(world)]
    counters = local_counters.clone()
    dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return torch.cat(parts, dim=0), counters
