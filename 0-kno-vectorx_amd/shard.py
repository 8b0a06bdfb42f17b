"""Multi-GPU sharding of independent proofs.

Proofs are independent units (each has its own 80-byte input and witness), so rank r of
`world` proves inputs {i : i mod world == r} with no data-path collective; the only exchange
is one gather of the fixed-capacity result blobs to rank 0 (RCCL over xGMI on GPUs, gloo in
the CPU tests).  The reference has no counterpart: it fans map/reduce jobs out over rayon
threads or an HTTP proof service (circuits/builder/subchain_verification.rs:72-79).
"""
import numpy as np


def assign_inputs(n_inputs, rank, world):
    """Indices of the inputs rank `rank` proves (round-robin, disjoint, covering)."""
    return list(range(rank, n_inputs, world))


def gather_blobs(blob, dist=None, device=None):
    """Gather one fixed-size uint8 blob per rank to rank 0.

    `dist` is torch.distributed (initialised) or None for a single process.  Returns the list
    of blobs (np.uint8 arrays, rank order) on rank 0 and None elsewhere.
    """
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    if dist is None or dist.get_world_size() == 1:
        return [blob]
    import torch

    mine = torch.from_numpy(blob.copy())
    if device is not None:
        mine = mine.to(device)
    # sizes must agree: a mismatch would hang the collective, so check it first
    sz = torch.tensor([mine.numel()], dtype=torch.int64, device=mine.device)
    lo, hi = sz.clone(), sz.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if int(lo.item()) != int(hi.item()):
        raise ValueError(f"blob sizes differ across ranks: {int(lo.item())}..{int(hi.item())}")
    rank, world = dist.get_rank(), dist.get_world_size()
    out = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, out, dst=0)
    return [t.cpu().numpy() for t in out] if rank == 0 else None
