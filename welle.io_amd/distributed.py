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


class FibGatherer:
    """gather_fibs with everything allocated ONCE (bench.py's N > 1 line times this inside its step): the rank's send buffer, rank 0's
    receive list and rank 0's page-locked landing area for all ranks' FIBs + CRC flags.  Per step: two device copies into the send
    buffer, ONE collective (dist.gather to rank 0), and on rank 0 one asynchronous copy per rank into the page-locked tensor and one
    synchronisation; the arrays handed out are views of that tensor (valid until the next call).
    shape_fib = (B_local, F, 12, 32), shape_ok = (B_local, F, 12); device: where the collective runs ("cuda:i" for RCCL, "cpu" for gloo)."""

    def __init__(self, dist, rank, world, shape_fib, shape_ok, device):
        import torch
        self.dist, self.rank, self.world = dist, rank, world
        self.shape_fib, self.shape_ok = tuple(shape_fib), tuple(shape_ok)
        self.nf, self.no = int(np.prod(shape_fib)), int(np.prod(shape_ok))
        self.device = torch.device(device)
        n = self.nf + self.no
        self.send = torch.empty(n, dtype=torch.uint8, device=self.device)
        self.recv = [torch.empty(n, dtype=torch.uint8, device=self.device) for _ in range(world)] if rank == 0 else None
        self.host = None
        if rank == 0:
            self.host = torch.empty((world, n), dtype=torch.uint8, pin_memory=self.device.type == "cuda")
            self.views = [(self.host[r, :self.nf].numpy().reshape(self.shape_fib), self.host[r, self.nf:].numpy().reshape(self.shape_ok)) for r in range(world)]

    def gather(self, fib, ok):
        """fib / ok: torch tensors (device buffers of the library, or host tensors for gloo) or numpy arrays of the shapes given at
        construction -> rank 0: list of (fib, ok) numpy views per rank, in rank order; other ranks: None"""
        import torch
        if not torch.is_tensor(fib):
            fib, ok = torch.from_numpy(np.ascontiguousarray(fib)), torch.from_numpy(np.ascontiguousarray(ok))
        assert fib.numel() == self.nf and ok.numel() == self.no, (tuple(fib.shape), tuple(ok.shape), self.shape_fib, self.shape_ok)
        self.send[:self.nf].copy_(fib.reshape(-1), non_blocking=True)
        self.send[self.nf:].copy_(ok.reshape(-1), non_blocking=True)
        self.dist.gather(self.send, self.recv, dst=0)
        if self.rank != 0:
            return None
        for r in range(self.world):
            self.host[r].copy_(self.recv[r], non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return self.views
