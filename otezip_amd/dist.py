"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

Archive entries are independent (every codec re-initialises per entry, ref
src/lib/otezip.c:503-522), so the path shards with NO data-path collective: each rank owns a
contiguous run of entries.  RCCL carries only the work queue and control traffic:
  * broadcast_table: rank 0 parses the central directory once and broadcasts the descriptor
    table (32 bytes per entry: 320 KB for 10 000 entries -- nothing next to 7 x 153 GB/s of xGMI);
  * partition_by_bytes: every rank derives the same contiguous, byte-balanced split from it;
  * the barrier around the timed region, the max-over-ranks of the elapsed time and the sum of
    the per-rank counters.
(Inside one process the C library deals the windows of an archive to several devices itself:
OTEZIP_GPU_DEVICES, otezip_amd/csrc/zip_api.c.)"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, device=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (no-op for a single process)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_total, world, rank):
    """Contiguous, balanced [lo, hi) of n_total entries for `rank` (sizes differ by at most 1)."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def partition_by_bytes(costs, world):
    """Contiguous [lo, hi) per rank with balanced total cost (costs[i] = comp_size + uncomp_size of
    entry i: the bytes the entry moves).  Contiguous runs keep each GPU's ingest a few large spans
    of the archive.  Every rank computes the same answer from the same table."""
    import numpy as np
    c = np.asarray(costs, dtype=np.float64)
    n = len(c)
    if n == 0:
        return [(0, 0)] * world
    cum = np.cumsum(c)
    total = float(cum[-1])
    cuts = [0]
    for r in range(1, world):
        # first index whose cumulative cost reaches r/world of the total, never behind the previous cut
        k = int(np.searchsorted(cum, total * r / world, side="left")) + 1 if total > 0 else n * r // world
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def broadcast_table(table, src=0, device="cpu"):
    """Rank `src` passes an int64 numpy array [n, k] (the descriptor table); every rank returns it.
    Two broadcasts: the shape, then the rows."""
    import numpy as np
    if not dist.is_initialized():
        return np.asarray(table, dtype=np.int64)
    rank = dist.get_rank()
    if rank == src:
        t = torch.as_tensor(np.ascontiguousarray(table, dtype=np.int64))
        shape = torch.tensor(list(t.shape) + [0] * (2 - t.dim()), dtype=torch.int64)
    else:
        t = None
        shape = torch.zeros(2, dtype=torch.int64)
    shape = shape.to(device)
    dist.broadcast(shape, src=src)
    n, k = int(shape[0].item()), int(shape[1].item())
    buf = t.to(device) if rank == src else torch.empty((n, k), dtype=torch.int64, device=device)
    dist.broadcast(buf, src=src)
    return buf.cpu().numpy()


def barrier():
    if dist.is_initialized():
        dist.barrier()


def reduce_max(x, device="cpu"):
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(x, device="cpu"):
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def finish():
    if dist.is_initialized():
        dist.destroy_process_group()
