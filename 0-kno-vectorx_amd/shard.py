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


class RcclComm:
    """A raw ncclComm_t (one rank per GPU) for the C-ABI gather `vx_gather_proofs` (include/vx.h), made with the RCCL copy
    this process ALREADY has mapped (the one torch.distributed's nccl backend loaded), else the one PyTorch ships, and only
    then a system librccl -- two different RCCL builds making communicators on one HIP runtime is what this order avoids.
    The unique id travels over the already initialised torch.distributed group.  torch must have initialised the HIP runtime
    on `device` first (RCCL wants the runtime it was built with).
    ncclCommInitRank is a collective: if one rank fails before it, the others block inside it.  The call therefore runs under
    a watchdog (VX_RCCL_INIT_TIMEOUT seconds, default 180): a rank still inside after that prints why and exits with code 3
    instead of hanging the job; bench.py's agreement check (all_reduce of an ok flag) covers the ranks that returned."""

    def __init__(self, dist, device):
        import ctypes as C
        import os

        import torch

        self.handle, self._lib = None, None
        mapped = []
        try:
            with open("/proc/self/maps") as f:
                mapped = sorted({ln.split()[-1] for ln in f if ln.split() and "librccl.so" in ln.split()[-1]})
        except OSError:
            pass
        for name in list(mapped) + [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"]:
            try:
                self._lib = C.CDLL(name)
                self.lib_path = name
                break
            except OSError:
                pass
        if self._lib is None:
            raise OSError("no RCCL library to make a communicator with")

        class Uid(C.Structure):
            _fields_ = [("b", C.c_char * 128)]

        rank, world = dist.get_rank(), dist.get_world_size()
        u = Uid()
        # rank 0 always broadcasts (None on failure): the other ranks must not be left waiting in the collective
        got_id = rank != 0 or self._lib.ncclGetUniqueId(C.byref(u)) == 0
        box = [(bytes(u) if got_id else None) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise OSError("ncclGetUniqueId failed on rank 0")
        C.memmove(C.byref(u), box[0], 128)
        torch.cuda.set_device(device)
        comm = C.c_void_p()
        self._lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
        import sys
        import threading

        res = []
        # the RCCL build PyTorch ships prints a version banner on STDOUT at the first communicator of a process: stdout carries the
        # bench's one JSON line, so file descriptor 1 points at stderr while the call runs
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            th = threading.Thread(target=lambda: res.append(self._lib.ncclCommInitRank(C.byref(comm), world, u, rank)), daemon=True)
            th.start()
            th.join(float(os.environ.get("VX_RCCL_INIT_TIMEOUT", "180")))
        finally:
            try:
                C.CDLL(None).fflush(None)  # the banner sits in C stdio's buffer (stdout is a pipe or a file: fully buffered)
            except OSError:
                pass
            os.dup2(saved, 1)
            os.close(saved)
        if th.is_alive():
            print(f"RcclComm: rank {rank} of {world} is still inside ncclCommInitRank -- another rank did not join; giving up", file=sys.stderr, flush=True)
            os._exit(3)
        rc = res[0]
        if rc != 0:
            raise OSError(f"ncclCommInitRank failed ({rc})")
        self.handle, self.world = comm.value, world

    def close(self):
        if self.handle:
            import ctypes as C

            self._lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self._lib.ncclCommDestroy(C.c_void_p(self.handle))
            self.handle = None


def gather_blobs_abi(ctx, comm, blob):
    """The same exchange through the exported C-ABI collective: all-gather of equal-length u64 blobs over RCCL on the
    ctx stream.  Returns the list of blobs (rank order) on EVERY rank."""
    words = np.ascontiguousarray(blob).view(np.uint8)
    if words.size % 8:
        words = np.concatenate([words, np.zeros(8 - words.size % 8, dtype=np.uint8)])
    got = ctx.gather_proofs(comm.handle, comm.world, words.view(np.uint64))
    return [got[r].view(np.uint8)[: np.asarray(blob).nbytes] for r in range(comm.world)]


# ---- intra-proof sharding: ONE proof, its tables spread over the ranks ------------------------------
# The hash-chain table of a header_range proof is split into map segments (the reference's MapReduce jobs,
# circuits/builder/subchain_verification.rs:72-79); segments and the four small tables are independent STARK tables on one
# logUp bus, table t (bus order) proven by rank t mod world.  The ranks meet ONCE per proof: after every table's trace is
# committed they all-reduce a small array of (public inputs, trace cap) slots -- the shared lookup challenges are a transcript of
# all of them -- and at the end rank 0 gathers the partial blobs and merges them (vx_header_range_merge).
def exchange_over(dist, device=None):
    """The all-reduce vx_header_range_prove_ex asks for (element-wise wrapping sum of uint64 words) on torch.distributed.
    It runs on one of the prover's host threads while the caller's thread is inside the C call."""
    import torch

    def exchange(words):
        t = torch.from_numpy(np.ascontiguousarray(words).view(np.int64).copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # int64 wrap-around = uint64 wrap-around; only one rank fills a slot anyway
        return t.cpu().numpy().view(np.uint64)

    return exchange


def prove_header_range_sharded(ctx, dist, headers_buf, stride, sizes, max_headers, trusted_block, trusted_hash, target_block, cfg=None, just=None,
                               n_segments=8, device=None, lib=None):
    """Every rank calls this with the SAME input (resident in ITS GPU's memory); returns (out96, blob) on rank 0 -- the blob is
    byte for byte what one prover makes with the same segment count -- and (out96, None) elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    out96, part = ctx.header_range_prove(headers_buf, stride, sizes, max_headers, trusted_block, trusted_hash, target_block, cfg, just=just, n_segments=n_segments,
                                         shard=(rank, world, exchange_over(dist, device)))
    return out96, gather_and_merge(part, dist, device=device, lib=lib)


def gather_and_merge(part, dist, device=None, lib=None):
    """Partial blobs of the shards (different lengths) -> the merged blob on rank 0, None elsewhere."""
    import torch

    part = np.ascontiguousarray(part, dtype=np.uint64)
    mx = torch.tensor([part.size], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    padded = np.zeros(int(mx.item()) + 1, dtype=np.uint64)
    padded[0] = part.size
    padded[1: 1 + part.size] = part
    got = gather_blobs(padded.view(np.uint8), dist, device=device)
    if dist.get_rank() != 0:
        return None
    blobs = [g.view(np.uint64) for g in got]
    if lib is None:
        from . import lib as lib_mod

        lib = lib_mod
    return lib.merge_blobs([b[1: 1 + int(b[0])] for b in blobs])
