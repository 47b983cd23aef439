"""Multi-GPU batched query: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  SURVEY.md 8(e): the query batch shards naturally --
the index (text + SA) is replicated on every GPU by one RCCL broadcast per tensor from the building
rank, ONE global batch is split into contiguous slices, every rank searches its slice with no
data-path collective, and the 8-byte (first,last) results are all-gathered.  Construction stays on
one GPU.  torch is plumbing here (device buffers + collectives); the search itself is the C ABI.

Everything below works on torch tensors that live where the process group's backend wants them
(HBM for nccl, host memory for gloo): bench.py --gpus N and tests/test_dist_cpu.py run the same
functions, only the `search` callable differs (DeviceIndex.query_batch_device / the oracle).
"""
import numpy as np
import torch
import torch.distributed as dist

PAIR_DTYPE = np.dtype([("first", "<u4"), ("second", "<u4")])


def shard_bounds(total, world_size, rank):
    """Contiguous, balanced slice [lo, hi) of `total` items for `rank`."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slot_count(total, world_size):
    """Result slots per rank in the gathered table: the largest slice (slices differ by at most one)."""
    return (int(total) + int(world_size) - 1) // int(world_size)


class _DevicePtr:
    """A raw device pointer as a __cuda_array_interface__ object (torch.as_tensor makes a view of it)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_view(ptr, count, dtype, device):
    """Tensor VIEW of `count` elements at device address `ptr` (the index's own text / SA buffers: the
    broadcast then reads them where they are -- no staging copy through the host)."""
    typestr = {torch.uint8: "|u1", torch.int32: "<i4", torch.int64: "<i8"}[dtype]
    return torch.as_tensor(_DevicePtr(ptr, count, typestr), device=device)


def broadcast_index(text_t, sa_t, src=0, group=None):
    """Replicate the index tensors (uint8 text, int32-viewed SA) from `src` to every rank.
    xGMI is point-to-point, so one large broadcast per tensor (RCCL pipelines it over the
    links) -- never per-chunk Python loops."""
    dist.broadcast(text_t, src=src, group=group)
    dist.broadcast(sa_t, src=src, group=group)
    return text_t, sa_t


class ShardedBatch:
    """This rank's slice of ONE global batch of `q` patterns + the buffers of the gather.

    patterns / offsets: host numpy arrays of the slice [lo, hi) = shard_bounds(q, world, rank)
    (offsets relative to the slice, uint64[hi - lo + 1]); they are moved to `device` once.
    out:      int32[2 * slots]  -- (first, second) pairs of the slice, padded to the common slot count
    gathered: int32[world * 2 * slots] -- every rank's `out`, rank-major (all_gather_into_tensor)
    """

    def __init__(self, patterns, offsets, q, world_size, rank, device):
        self.q, self.world, self.rank = int(q), int(world_size), int(rank)
        self.lo, self.hi = shard_bounds(q, world_size, rank)
        assert offsets.size == self.hi - self.lo + 1
        self.slots = slot_count(q, world_size)
        pad = np.zeros(64, np.uint8)   # the device search reads a pattern's last partial word whole (sa_hip.h)
        self.pat = torch.from_numpy(np.concatenate([np.ascontiguousarray(patterns, dtype=np.uint8), pad])).to(device)
        self.off = torch.from_numpy(np.ascontiguousarray(offsets, dtype=np.uint64).view(np.int64)).to(device)
        self.out = torch.zeros(2 * self.slots, dtype=torch.int32, device=device)
        self.gathered = torch.empty(self.world * 2 * self.slots, dtype=torch.int32, device=device)

    @property
    def q_local(self):
        return self.hi - self.lo

    def step(self, search, group=None):
        """One pass of the multi-GPU hot path: search this rank's slice, gather everybody's ranges.
        search(pat_tensor, off_tensor, q_local, out_tensor) fills out[0 : 2 * q_local] and must have
        completed (stream-synchronised) when it returns: the collective runs on torch's stream."""
        if self.q_local:
            search(self.pat, self.off, self.q_local, self.out)
        if self.world > 1 or dist.is_initialized():
            dist.all_gather_into_tensor(self.gathered, self.out, group=group)
        else:
            self.gathered.copy_(self.out)
        return self.gathered

    def results(self):
        """The gathered table as one structured (first, second) array of the whole batch, in batch order."""
        g = self.gathered.cpu().numpy().view(np.uint32).reshape(self.world, self.slots, 2)
        out = np.zeros(self.q, dtype=PAIR_DTYPE)
        for r in range(self.world):
            a, b = shard_bounds(self.q, self.world, r)
            out["first"][a:b] = g[r, :b - a, 0]
            out["second"][a:b] = g[r, :b - a, 1]
        return out
