"""Multi-GPU layer of the hot path: ensembles are independent streams, so the batch is sharded by ensemble (one
process per GPU, no exchange while decoding) and only the decoded FIBs + CRC flags are gathered to rank 0 -- the one
collective the path has (torch.distributed: backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests)."""
import numpy as np


def shard_range(n_items, rank, world):
    """contiguous block of ensembles owned by `rank` (first ranks take the remainder)"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_fibs(dist, fib, ok, rank, world, device=None):
    """fib [B_local, F, 12, 32] uint8, ok [B_local, F, 12] uint8 on every rank -> rank 0 gets the lists of all ranks
    (same shard order), other ranks get None.  Shards must have equal shapes (weak scaling: B per GPU fixed)."""
    import torch
    t = torch.from_numpy(np.concatenate([fib.reshape(-1), ok.reshape(-1)]))
    if device is not None:
        t = t.to(device)
    gl = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gl, dst=0)
    if rank != 0:
        return None
    nf = fib.size
    out = []
    for g in gl:
        a = g.cpu().numpy()
        out.append((a[:nf].reshape(fib.shape), a[nf:].reshape(ok.shape)))
    return out
