"""One rank of tests/test_gpu_dist.py::test_replicate_index_two_ranks_one_gpu: started as a FRESH process (no GPU call before
torch.distributed is up), backend gloo, both ranks on device 0 (RCCL refuses two ranks on one GPU; gloo moves host copies:
distributed.host_staged_broadcast).  Rank 0 builds and replicates, rank 1 RECEIVES -- layout decoded from the broadcast,
replica_reserve, receive, replica_commit -- and both search their slice of one global batch (bench.py's config-4 step at
world size 2).  Rank 1 additionally answers the whole batch from its replica.  Results go to <out>/r<rank>.npz; the parent
checks them against the oracle.
    python tests/dist_gpu_worker.py <out_dir> <n> <q> <mode> <chunks>     (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, n, q, mode, chunks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from suffixarray_amd import _capi, synth
    from suffixarray_amd.distributed import ShardedBatch, host_staged_broadcast, replicate_index, shard_bounds
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    m = 16
    text = synth.d1_uniform27(n)          # every rank draws its patterns from the same text; only rank 0 indexes it
    idx = _capi.DeviceIndex(n, 0)
    if rank == 0:
        idx.build(text)
    moved = replicate_index(idx if rank == 0 else None, None if rank == 0 else idx, dev, src=0, transport=host_staged_broadcast)
    lo, hi = shard_bounds(q, world, rank)
    buf, off = synth.query_batch(text, q, m, seed=0, lo=lo, hi=hi)
    batch = ShardedBatch(buf, off, q, world, rank, dev, chunks=chunks, mode=mode,
                         search_stream=torch.cuda.ExternalStream(idx.stream, device=dev), stage_host=True)

    def search(pat_t, off_t, start, count, out_t):
        idx.query_batch_device_fixed(pat_t.data_ptr() + m * start, m, count, out_t.data_ptr())

    for _ in range(2):
        batch.step(search)
    got = batch.results()
    res = {"moved": np.array([moved], dtype=np.int64)}
    if got is not None:
        res.update(first=got["first"], second=got["second"])
    if rank == 0:
        res["sa"] = idx.sa_u32()
    else:
        fb, fo = synth.query_batch(text, q, m, seed=0)
        whole = idx.query_batch((fb, fo))              # the replica alone, whole batch
        res.update(whole_first=whole["first"], whole_second=whole["second"], replica_sa=idx.sa_u32(),
                   freq=idx.freq(), n=np.array([idx.n], dtype=np.int64))
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), **res)
    idx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
