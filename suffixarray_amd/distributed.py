"""Multi-GPU batched query: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  SURVEY.md 8(e): the query batch shards naturally --
the index is replicated on every GPU by RCCL broadcasts from the building rank, ONE global batch is
split into contiguous slices, every rank searches its slice with no data-path collective, and the
8-byte (first,last) results are gathered.  Construction stays on one GPU.  torch is plumbing here
(device buffers, streams, collectives); the search itself is the C ABI (the same replication without
torch: sa_hip_comm_* in include/sa_hip.h).

Everything below works on torch tensors that live where the process group's backend wants them
(HBM for nccl, host memory for gloo): bench.py --gpus N, tests/test_gpu_dist.py and tests/test_dist_cpu.py
run the same functions, only the `search` callable differs (DeviceIndex.query_batch_device* / the oracle).

Streams.  The index launches on its own (non-blocking) HIP stream, torch's collectives are ordered after
torch's current stream.  ShardedBatch.step orders the two with events only -- the host never waits:
    search chunk k   (index stream)   waits for: the gather of chunk k of the PREVIOUS step (it reads out[k])
    gather chunk k   (torch's stream) waits for: search chunk k
so the search of chunk k + 1 runs while chunk k's ranges travel.  results() synchronises.
"""
import ctypes as C

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
    """Tensor VIEW of `count` elements at device address `ptr` (the index's own buffers: a broadcast then
    reads / writes them where they are -- no staging copy)."""
    typestr = {torch.uint8: "|u1", torch.int32: "<i4", torch.int64: "<i8"}[dtype]
    return torch.as_tensor(_DevicePtr(ptr, count, typestr), device=device)


def broadcast_index(text_t, sa_t, src=0, group=None):
    """Replicate plain (uint8 text, int32-viewed SA) tensors from `src` to every rank.  Only tests/test_dist_cpu.py uses
    this form (its searcher is the oracle, which needs nothing but text and SA); bench.py --gpus N and the GPU tests
    replicate a built index with replicate_index below.  xGMI is point-to-point, so one large broadcast per
    tensor (RCCL pipelines it over the links) -- never per-chunk Python loops."""
    dist.broadcast(text_t, src=src, group=group)
    dist.broadcast(sa_t, src=src, group=group)
    return text_t, sa_t


def host_staged_broadcast(t, src=0, group=None):
    """Transport for replicate_index on a backend that only moves host memory (gloo): the device tensor goes through a
    host copy on both sides.  tests/test_gpu_dist.py uses it to run two ranks on ONE GPU, which RCCL refuses."""
    h = t.cpu() if dist.get_rank(group) == src else torch.empty(t.shape, dtype=t.dtype)
    dist.broadcast(h, src=src, group=group)
    if dist.get_rank(group) != src:
        t.copy_(h)


def replicate_index(src_idx, dst_idx, device, src=0, group=None, transport=None):
    """Replicate a built device index WITHOUT rebuilding anything on the receiving side (sa_hip_index_replica_*):
    the layout (a small struct) and then text, suffix array, sorted key array and bucket directory are broadcast
    straight out of the builder's buffers into buffers the replica has reserved; commit range-checks the SA.

    src_idx: the built DeviceIndex on rank `src` (None elsewhere).
    dst_idx: the DeviceIndex that becomes a replica (None on a rank that keeps searching src_idx).  On rank `src`
             itself a dst_idx is filled by a device-to-device copy (world size 1 then runs every line a receiving rank runs).
    transport: callable(tensor, src, group) that makes `tensor` (a device tensor, in place) equal to rank src's on every rank;
               default dist.broadcast (RCCL).
    Returns the bytes replicated (the broadcast seconds are the caller's to time)."""
    bcast = transport if transport is not None else (lambda t, src, group: dist.broadcast(t, src=src, group=group))
    from ._capi import ReplicaLayout
    rank = dist.get_rank(group)
    lay = ReplicaLayout()
    lay_t = torch.zeros(C.sizeof(ReplicaLayout), dtype=torch.uint8, device=device)
    if rank == src:
        lay = src_idx.replica_layout()
        lay_t.copy_(torch.frombuffer(bytearray(bytes(lay)), dtype=torch.uint8))
    bcast(lay_t, src, group)
    if rank != src:
        lay = ReplicaLayout.from_buffer_copy(lay_t.cpu().numpy().tobytes())
    sbufs = src_idx.replica_buffers().items() if rank == src else None     # synchronises the builder's stream
    dbufs = dst_idx.replica_reserve(lay).items() if dst_idx is not None else None
    total = 0
    for i in range(4):
        ptr, nbytes = sbufs[i] if rank == src else dbufs[i]
        if nbytes == 0:
            continue
        t = device_view(ptr, nbytes, torch.uint8, device)
        bcast(t, src, group)
        if rank == src and dbufs is not None:
            device_view(dbufs[i][0], nbytes, torch.uint8, device).copy_(t)
        total += nbytes
    if device.type == "cuda":
        torch.cuda.current_stream(device).synchronize()    # the buffers are complete before the index looks at them
    if dst_idx is not None:
        dst_idx.replica_commit()
    return total


class ShardedBatch:
    """This rank's slice of ONE global batch of `q` patterns + the buffers of the gather.

    patterns / offsets: host numpy arrays of the slice [lo, hi) = shard_bounds(q, world, rank)
    (offsets relative to the slice, uint64[hi - lo + 1]); they are moved to `device` once.
    The slice is searched in `chunks` pieces of cslots = ceil(slots / chunks) patterns:
    out[c]:      int32[2 * cslots]          -- (first, second) pairs of chunk c, padded to the common slot count
    gathered[c]: int32[world][2 * cslots]   -- every rank's out[c] (all_gather_into_tensor; mode "gather_to_root":
                                               dist.gather, the table exists on rank 0 only)
    search_stream: the stream the `search` callable launches on (torch.cuda.ExternalStream of the index's own
    stream), or None when search() has completed on return (the CPU test; a searcher that synchronises).
    """

    def __init__(self, patterns, offsets, q, world_size, rank, device, chunks=1, mode="all_gather", search_stream=None,
                 stage_host=False):
        assert mode in ("all_gather", "gather_to_root")
        # stage_host: the collective moves host copies of the device buffers (a backend that only takes host memory: two gloo
        # ranks on one GPU in tests/test_gpu_dist.py); synchronises per chunk, so nothing overlaps.
        self.stage_host = bool(stage_host)
        self.q, self.world, self.rank = int(q), int(world_size), int(rank)
        self.lo, self.hi = shard_bounds(q, world_size, rank)
        assert offsets.size == self.hi - self.lo + 1
        self.slots = slot_count(q, world_size)
        self.chunks = max(1, min(int(chunks), max(self.slots, 1)))
        self.cslots = (self.slots + self.chunks - 1) // self.chunks
        self.mode = mode
        self.device = device
        self.search_stream = search_stream
        pad = np.zeros(64, np.uint8)   # the device search reads a pattern's last partial word whole (sa_hip.h)
        self.pat = torch.from_numpy(np.concatenate([np.ascontiguousarray(patterns, dtype=np.uint8), pad])).to(device)
        self.off = torch.from_numpy(np.ascontiguousarray(offsets, dtype=np.uint64).view(np.int64)).to(device)
        # every slot is written by a search or is padding nobody reads: no fill kernel that could race with the first search
        self.out = torch.empty(self.chunks, 2 * self.cslots, dtype=torch.int32, device=device)
        self.root = (mode == "all_gather") or rank == 0
        self.gathered = (torch.empty(self.chunks, self.world, 2 * self.cslots, dtype=torch.int32, device=device)
                         if self.root else None)
        cuda = device.type == "cuda"
        self._done = [torch.cuda.Event() for _ in range(self.chunks)] if cuda and search_stream is not None else None
        self._free = [torch.cuda.Event() for _ in range(self.chunks)] if cuda and search_stream is not None else None
        if cuda:
            torch.cuda.current_stream(device).synchronize()   # the uploads above are done before any other stream reads them

    @property
    def q_local(self):
        return self.hi - self.lo

    def chunk_bounds(self, c):
        """patterns [a, b) of this rank's slice that chunk c searches (relative to the slice)"""
        a = min(c * self.cslots, self.q_local)
        return a, min(a + self.cslots, self.q_local)

    def step(self, search, group=None):
        """One pass of the multi-GPU hot path: search this rank's slice chunk by chunk, gather everybody's ranges.
        search(pat_tensor, off_tensor, start, count, out_tensor): patterns [start, start + count) of the slice ->
        out_tensor[0 : 2 * count]; it launches on `search_stream` (asynchronous) or has completed on return."""
        for c in range(self.chunks):
            a, b = self.chunk_bounds(c)
            if self._free is not None:
                self.search_stream.wait_event(self._free[c])      # the previous step's gather has read out[c]
            if b > a:
                search(self.pat, self.off, a, b - a, self.out[c])
            if self._done is not None:
                self._done[c].record(self.search_stream)
                torch.cuda.current_stream(self.device).wait_event(self._done[c])
            if self.stage_host:
                torch.cuda.synchronize(self.device)
                mine = self.out[c].cpu()
                if self.mode == "all_gather":
                    table = torch.empty(self.world, 2 * self.cslots, dtype=torch.int32)
                    dist.all_gather_into_tensor(table.view(-1), mine, group=group)
                    self.gathered[c].copy_(table)
                else:
                    parts = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == 0 else None
                    dist.gather(mine, parts, dst=0, group=group)
                    if self.rank == 0:
                        self.gathered[c].copy_(torch.stack(parts))
            elif self.world > 1 or dist.is_initialized():
                if self.mode == "all_gather":
                    dist.all_gather_into_tensor(self.gathered[c].view(-1), self.out[c], group=group)
                else:
                    dist.gather(self.out[c], list(self.gathered[c].unbind(0)) if self.rank == 0 else None, dst=0, group=group)
            else:
                self.gathered[c, 0].copy_(self.out[c])
            if self._free is not None:
                self._free[c].record(torch.cuda.current_stream(self.device))
        return self.gathered

    def results(self):
        """The gathered table as one structured (first, second) array of the whole batch, in batch order
        (mode "gather_to_root": on rank 0; None elsewhere).  Waits for everything in flight."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if self.gathered is None:
            return None
        g = self.gathered.cpu().numpy().view(np.uint32).reshape(self.chunks, self.world, self.cslots, 2)
        out = np.zeros(self.q, dtype=PAIR_DTYPE)
        for r in range(self.world):
            lo, hi = shard_bounds(self.q, self.world, r)
            for c in range(self.chunks):
                a = min(c * self.cslots, hi - lo)
                b = min(a + self.cslots, hi - lo)
                out["first"][lo + a:lo + b] = g[c, r, :b - a, 0]
                out["second"][lo + a:lo + b] = g[c, r, :b - a, 1]
        return out
