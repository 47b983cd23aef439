"""GPU: BASELINE config 4's code path on one MI355X -- RCCL initialisation, index replication without rebuilding
(sa_hip_index_replica_*), the sharded, pipelined batch (ShardedBatch: search chunk k + 1 while chunk k is gathered) and
the gather -- run at world size 1 in a FRESH child process (bench.py --exercise-dist: the process initialises torch,
RCCL and the GPU in the order a torch.distributed.run rank does), and checked here against the ORACLE: the gathered
ranges against oracle.query_batch over the downloaded suffix array, the suffix array against the oracle's SA-IS.
The replica entry points are additionally driven directly, in this process, with device-to-device copies as the
transport.  (World size 2 of the same functions: tests/test_dist_cpu.py on gloo.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from suffixarray_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,chunks,offsets_api", [("all_gather", 4, False), ("gather_to_root", 3, True)])
def test_config4_path_world1_child_process(gpu, oracle, tmp_path, mode, chunks, offsets_api):
    n, q, m = 20_000_000, 400_003, 16
    dump = str(tmp_path / "ranges.npz")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--exercise-dist", "--chars", str(n), "--queries-global", str(q),
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--dist-mode", mode, "--dist-chunks", str(chunks), "--dist-min-chunk", "1", "--dump", dump]
    if offsets_api:
        cmd.append("--offsets-api")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["gate"]["ok"] and line["gate"]["verify_violations"] == 0, line["gate"]
    assert line["replicate_bytes"] >= 9 * n          # text + SA + narrow keys + directory all travelled
    assert line["one_gpu_same_batch"]["queries_per_s"] > 0   # the 1-GPU reference for `value` rides in the same line
    # ... and against the oracle, not against the builder's own answers
    d = np.load(dump)
    text = synth.d1_uniform27(n)
    sa = oracle.sais(text).astype(np.uint32)
    assert np.array_equal(d["sa"], sa)
    fb, fo = synth.query_batch(text, q, m, seed=0)
    exp = oracle.query_batch(text, sa, 0xFFFFFFFF, (fb, fo))
    assert np.array_equal(d["first"], exp["first"]) and np.array_equal(d["second"], exp["second"])
    hits = ((exp["second"].astype(np.int64) - exp["first"].astype(np.int64) + 1) & 0xFFFFFFFF) > 0
    assert 0.05 < hits.mean() < 0.95                 # both sub-populations present
    print("config-4 path, world size 1 (%s, %d chunks): %.2f G queries/s end to end, %.2f in the kernels; replicate %.1f ms for %.0f MB" % (
        mode, chunks, line["value"] / 1e9, line["search_only_queries_per_s"] / 1e9, line["replicate_ms"], line["replicate_bytes"] / 1e6))


@pytest.mark.parametrize("mode,chunks", [("all_gather", 2), ("gather_to_root", 1)])
def test_replicate_index_two_ranks_one_gpu(gpu, oracle, tmp_path, mode, chunks):
    """World size 2 on ONE MI355X: two fresh child processes on device 0, backend gloo with host-staged copies as the transport
    (RCCL refuses two ranks on one GPU).  Rank 1 runs the RECEIVING side of replicate_index -- the layout decoded from the
    broadcast, replica_reserve, the four buffers received, replica_commit -- from a layout that really crossed a process
    boundary, then answers its slice (and, alone, the whole batch).  Everything is compared with the oracle here."""
    n, q, m = 20_000_000, 300_001, 16
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path), str(n), str(q), mode, str(chunks)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for x in procs:
                x.kill()
            raise
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    text = synth.d1_uniform27(n)
    sa = oracle.sais(text).astype(np.uint32)
    assert np.array_equal(r0["sa"], sa) and np.array_equal(r1["replica_sa"], sa)
    assert int(r1["n"][0]) == n and np.array_equal(r1["freq"], np.bincount(text, minlength=256))
    assert int(r0["moved"][0]) == int(r1["moved"][0]) >= 9 * n
    fb, fo = synth.query_batch(text, q, m, seed=0)
    exp = oracle.query_batch(text, sa, 0xFFFFFFFF, (fb, fo))
    assert np.array_equal(r1["whole_first"], exp["first"]) and np.array_equal(r1["whole_second"], exp["second"])
    assert np.array_equal(r0["first"], exp["first"]) and np.array_equal(r0["second"], exp["second"])   # the gathered table on rank 0
    if mode == "all_gather":
        assert np.array_equal(r1["first"], exp["first"]) and np.array_equal(r1["second"], exp["second"])
    else:
        assert "first" not in r1.files


def _replicate_locally(gpu, src, n_max):
    """The replica entry points with torch device-to-device copies as the transport (what RCCL does between GPUs)."""
    import torch
    from suffixarray_amd.distributed import device_view
    dev = torch.device("cuda", 0)
    dst = gpu.DeviceIndex(n_max, 0)
    lay = src.replica_layout()
    sb = src.replica_buffers().items()
    db = dst.replica_reserve(lay).items()
    for (sp, nb), (dp, nb2) in zip(sb, db):
        assert nb == nb2
        if nb:
            device_view(dp, nb, torch.uint8, dev).copy_(device_view(sp, nb, torch.uint8, dev))
    torch.cuda.synchronize()
    return dst, lay, db


def test_replica_answers_like_the_builder_and_the_oracle(gpu, oracle):
    """Narrow key array (near-random text), u64 key array (word text: 12-character keys), a truncated index and a tiny
    one: the replica gives the builder's and the oracle's ranges without having built or searched anything itself."""
    import cases
    rng = np.random.default_rng(5)
    runs = [("d1", synth.d1_uniform27(4_600_000), 0, 4), ("words", synth.d2_words(3_000_000), 0, 8),
            ("words_L20", synth.d2_words(1_000_000), 20, 8), ("tiny", np.frombuffer(b"abracadabra", np.uint8), 0, 8),
            ("one", np.frombuffer(b"z", np.uint8), 0, 0)]
    for name, text, L, key_bytes in runs:
        with gpu.DeviceIndex(text.size, 0) as src:
            src.build(text, L)
            dst, lay, _ = _replicate_locally(gpu, src, text.size)
            try:
                assert lay.n == text.size and lay.key_bytes == key_bytes, (name, lay.key_bytes)
                dst.replica_commit()
                pats = cases.query_patterns(text, 2000, rng)
                got = dst.query_batch(pats)
                assert np.array_equal(got, src.query_batch(pats)), name
                sa = src.sa_u32()
                assert np.array_equal(dst.sa_u32(), sa)
                assert np.array_equal(got, oracle.query_batch(text, sa, L if L else 0xFFFFFFFF, pats)), name
                assert np.array_equal(dst.freq(), src.freq()) and dst.max_suffix_length == L
            finally:
                dst.close()


def test_replica_commit_refuses_what_would_send_the_search_out_of_bounds(gpu):
    import torch
    from suffixarray_amd.distributed import device_view
    text = synth.d1_uniform27(4_300_000)
    dev = torch.device("cuda", 0)
    with gpu.DeviceIndex(text.size, 0) as src:
        src.build(text)
        dst, lay, db = _replicate_locally(gpu, src, text.size)
        device_view(db[1][0], text.size, torch.int32, dev)[12345] = -7            # an SA entry >= n
        torch.cuda.synchronize()
        with pytest.raises(gpu.SaHipError):
            dst.replica_commit()
        with pytest.raises(gpu.SaHipError):
            dst.query_batch([b"abc"])                                              # not searchable
        with pytest.raises(gpu.SaHipError):
            dst.replica_commit()                                                   # nothing reserved any more
        dst.close()
        dst, lay, db = _replicate_locally(gpu, src, text.size)
        device_view(db[3][0], int(lay.dir_entries), torch.int32, dev)[-1] = 17     # the directory's end marker
        torch.cuda.synchronize()
        with pytest.raises(gpu.SaHipError):
            dst.replica_commit()
        dst.close()
        with gpu.DeviceIndex(1000, 0) as small:
            with pytest.raises(gpu.SaHipError):
                small.replica_reserve(lay)                                         # beyond the handle's capacity


def test_comm_c_abi_world_size_1(gpu, oracle):
    """sa_hip_comm_*: RCCL through the C ABI, no torch.distributed -- unique id, communicator, index replication entry point
    (at world size 1 the root's part of it) and the gather of the range pairs on the index's own stream, ordered after the
    search without a host synchronisation in between."""
    import torch
    text = synth.d1_uniform27(4_400_000)
    q, m = 50_000, 12
    fb, fo = synth.query_batch(text, q, m, seed=3)
    dev = torch.device("cuda", 0)
    with gpu.Comm(gpu.Comm.unique_id(), 1, 0, 0) as comm, gpu.DeviceIndex(text.size, 0) as idx:
        assert comm.rank == 0 and comm.size == 1
        idx.build(text)
        moved = comm.replicate_index(idx, root=0)
        assert moved >= 9 * text.size
        pat = torch.from_numpy(np.concatenate([fb, np.zeros(64, np.uint8)])).to(dev)
        out = torch.full((2 * q,), -1, dtype=torch.int32, device=dev)
        rec = torch.full((2 * q,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        idx.query_batch_device_fixed(pat.data_ptr(), m, q, out.data_ptr())
        comm.allgather_ranges(idx, out.data_ptr(), q, rec.data_ptr())     # no sync in between: same stream
        idx.sync()
        got = rec.cpu().numpy().view(np.uint32).reshape(q, 2)
        exp = oracle.query_batch(text, idx.sa_u32(), 0xFFFFFFFF, (fb, fo))
        assert np.array_equal(got[:, 0], exp["first"]) and np.array_equal(got[:, 1], exp["second"])
        with pytest.raises(gpu.SaHipError):
            comm.replicate_index(idx, root=3)
