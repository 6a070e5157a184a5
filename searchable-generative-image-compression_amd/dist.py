"""Multi-GPU plumbing: one process per GPU, images sharded across ranks, ONE collective -- the
all-gather of the CLIP vectors for the FAISS index (the reference gathers through *.npy files on a shared
file system, compress.py:286,295-306).  Backend "nccl" is RCCL over xGMI on MI355X; the same code runs on
"gloo" for the CPU tests."""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """contiguous shard [lo, hi) of n items for `rank` (first n % world ranks get one extra)"""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_vectors(local, n_total, rank, world):
    """local: (n_local, D) fp32 rows of this rank's contiguous shard -> (n_total, D) on every rank.
    Shards differ by at most one row, so each rank pads to ceil(n/world) rows and the pad is dropped.
    Without a process group (plain single-process run) the shard is the whole; WITH one the collective runs even at
    world 1 -- a `torchrun --nproc-per-node 1` job takes exactly the code path of an 8-rank job."""
    if not (dist.is_available() and dist.is_initialized()):
        assert world == 1, "world > 1 needs an initialised process group"
        return local
    if n_total == 0:
        return local
    D = local.shape[1]
    per = (n_total + world - 1) // world
    buf = torch.zeros(per, D, dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out = torch.empty(world * per, D, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r * per:r * per + (hi - lo)])
    return torch.cat(parts, dim=0)
