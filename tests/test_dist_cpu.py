"""CPU: the N > 1 path of the batched query (suffixarray_amd/distributed.py: broadcast_index, shard_bounds,
ShardedBatch.step / results -- the functions bench.py --gpus N runs) with world_size 2 on the gloo backend.
The per-rank searcher is the oracle here (no GPU in this container); on the GPU box it is
DeviceIndex.query_batch_device."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, tmp, chunks=1, mode="all_gather"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle.oracle import Oracle
    from suffixarray_amd import synth
    from suffixarray_amd.distributed import ShardedBatch, broadcast_index, shard_bounds
    o = Oracle()
    n = 200_000
    m = 12
    dev = torch.device("cpu")
    # rank 0 "builds"; the index reaches the other rank only through the broadcast (bench.py: run_sharded)
    if rank == 0:
        text = synth.d1_uniform27(n)
        sa = o.sais(text).astype(np.int32)
        tx_t, sa_t = torch.from_numpy(text.copy()), torch.from_numpy(sa.copy())
    else:
        tx_t, sa_t = torch.zeros(n, dtype=torch.uint8), torch.zeros(n, dtype=torch.int32)
    broadcast_index(tx_t, sa_t, src=0)
    text = tx_t.numpy()
    sa = sa_t.numpy().astype(np.uint32)
    ref_text = synth.d1_uniform27(n)
    # ONE global batch; this rank generates and holds only its slice
    lo, hi = shard_bounds(q, world, rank)
    buf, off = synth.query_batch(ref_text, q, m, lo=lo, hi=hi)
    batch = ShardedBatch(buf, off, q, world, rank, dev, chunks=chunks, mode=mode)

    def search(pat_t, off_t, start, count, out_t):
        # the local searcher of the CPU test is the oracle; on the GPU it is sa_hip_query_batch_device[_fixed]
        pat = pat_t.numpy()
        offs = off_t.numpy().view(np.uint64)[start:start + count + 1]
        res = o.query_batch(text, sa, 0xFFFFFFFF, (pat[int(offs[0]):int(offs[-1])], offs - offs[0]), threads=1)
        out_t[:2 * count] = torch.from_numpy(res.view(np.uint32).view(np.int32).copy())

    for _ in range(2):   # the buffers are reused from step to step
        batch.step(search)
    got = batch.results()
    if got is None:      # gather_to_root: only rank 0 holds the table
        ok = rank != 0
    else:
        fb, fo = synth.query_batch(ref_text, q, m)
        exp = o.query_batch(ref_text, o.sais(ref_text).astype(np.uint32), 0xFFFFFFFF, (fb, fo), threads=1)
        ok = np.array_equal(got["first"], exp["first"]) and np.array_equal(got["second"], exp["second"])
    np.save(os.path.join(tmp, f"r{rank}.npy"), np.array([int(ok), lo, hi]))
    dist.destroy_process_group()


@pytest.mark.parametrize("q,chunks,mode", [(1001, 1, "all_gather"), (7, 1, "all_gather"), (1, 1, "all_gather"),
                                           (1001, 4, "all_gather"), (1003, 3, "gather_to_root"), (5, 4, "gather_to_root")])
def test_sharded_query_world2_gloo(tmp_path, q, chunks, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), q, str(tmp_path), chunks, mode), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{k}.npy") for k in range(world)]
    assert all(x[0] == 1 for x in r)
    assert r[0][1] == 0 and r[0][2] == r[1][1] and r[1][2] == q   # contiguous, complete slices


def test_shard_bounds_cover_everything():
    from suffixarray_amd.distributed import shard_bounds
    for total in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
