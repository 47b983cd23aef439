"""Multi-GPU batched query: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  SURVEY.md 8(e): the query batch shards naturally --
the index (text + SA) is replicated on every GPU by one RCCL broadcast from the building rank,
the batch is split into contiguous slices, every rank searches its slice with no data-path
collective, and the 8-byte (first,last) results are all-gathered.  Construction stays on one
GPU.  torch is plumbing here (device buffers + collectives); the search itself is the C ABI.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous, balanced slice [lo, hi) of `total` items for `rank`."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_index(text_t, sa_t, src=0, group=None):
    """Replicate the index tensors (uint8 text, int32-viewed SA) from `src` to every rank.
    xGMI is point-to-point, so one large broadcast per tensor (RCCL pipelines it over the
    links) -- never per-chunk Python loops."""
    dist.broadcast(text_t, src=src, group=group)
    dist.broadcast(sa_t, src=src, group=group)
    return text_t, sa_t


def sharded_query(local_query_fn, patterns, offsets, world_size, rank, device, group=None):
    """Run `local_query_fn(packed_u8, offsets_u64) -> structured (first, second) array` on this
    rank's slice of the batch and all-gather the results.  patterns/offsets: host numpy arrays
    describing the WHOLE batch (every rank holds them; only the slice is searched).
    Returns the full result array on every rank (numpy, PAIR dtype order first,second)."""
    q = offsets.size - 1
    lo, hi = shard_bounds(q, world_size, rank)
    loc_off = (offsets[lo:hi + 1] - offsets[lo]).astype(np.uint64)
    loc_pat = patterns[int(offsets[lo]):int(offsets[hi])]
    res = local_query_fn(np.ascontiguousarray(loc_pat), np.ascontiguousarray(loc_off))
    flat = np.ascontiguousarray(res).view(np.uint32).astype(np.int64).reshape(-1)  # (first, second) pairs
    # equal-sized all_gather: pad every slice to the largest one
    per = (q + world_size - 1) // world_size
    send = torch.zeros(2 * per, dtype=torch.int64, device=device)
    send[:flat.size] = torch.from_numpy(flat).to(device)
    recv = [torch.empty_like(send) for _ in range(world_size)]
    dist.all_gather(recv, send, group=group)
    out = np.zeros(q, dtype=np.dtype([("first", "<u4"), ("second", "<u4")]))
    for r in range(world_size):
        a, b = shard_bounds(q, world_size, r)
        part = recv[r][:2 * (b - a)].cpu().numpy().astype(np.uint32).reshape(-1, 2)
        out["first"][a:b] = part[:, 0]
        out["second"][a:b] = part[:, 1]
    return out
