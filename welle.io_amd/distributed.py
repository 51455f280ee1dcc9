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
    """fib [B_local, F, 12, 32] uint8, ok [B_local, F, 12] uint8 on every rank -> rank 0 gets the list of all ranks' (fib, ok) as
    host arrays (same shard order), other ranks get None.  Shards must have equal shapes (weak scaling: B per GPU fixed).
    fib / ok may be torch tensors that alias the library's HBM buffers (capi.DabPhy.fibs_device(): the RCCL gather then moves
    device memory to device memory and only rank 0's result crosses to its host) or numpy arrays (moved to `device` first, if given)."""
    import torch
    if not torch.is_tensor(fib):
        fib, ok = torch.from_numpy(np.ascontiguousarray(fib)), torch.from_numpy(np.ascontiguousarray(ok))
        if device is not None:
            fib, ok = fib.to(device), ok.to(device)
    t = torch.cat([fib.reshape(-1), ok.reshape(-1)])          # one message per rank and step (2 MB at 256 x 20 frames)
    gl = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gl, dst=0)
    if rank != 0:
        return None
    nf = fib.numel()
    out = []
    for g in gl:
        a = g.cpu().numpy()
        out.append((a[:nf].reshape(tuple(fib.shape)), a[nf:].reshape(tuple(ok.shape))))
    return out
