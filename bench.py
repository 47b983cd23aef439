#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric):
SA build chars/s (N = 1e9) + batched queries/s (Q = 1e6) on one MI355X, % of HBM roofline.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--chars CHARS] [--queries QUERIES]

A "step" = one device build of the suffix array of the N-char synthetic text D1 (uniform27,
SURVEY.md 8d; text resident in HBM before the timed region) + one batch of Q 16-byte queries
(50 % text windows, 50 % random; patterns resident in HBM).  N > 1 GPUs (launched by
torch.distributed.run, one rank per GPU): construction does not shard ("replicas only", every
rank builds its own replica); the query batch is weak-scaled (Q per GPU) and the results are
all-gathered over RCCL; the one-time RCCL broadcast of (text, SA) is timed separately.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # MI355X HBM3E peak, B/s (MI355X_MICROARCH.md)


def usable_cpus():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pd = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, q // pd))
        except Exception:
            pass
    return n


PASS_KERNELS = ["radix_onesweep_kernel<512, 0, false>", "radix_onesweep_kernel<512, 0, true>",
                "seg_onesweep_kernel<512, 24, false, true>", "seg_onesweep_kernel<512, 24, true, true>"]   # sa_hip_build_stats.pass_*


def pmc_traffic(n, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc run
    (profiles/pmc_onesweep.json; PMC passes cannot run inside this process).  Only reported for
    the workload and the kernel it was measured on."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "pmc_onesweep.json")))
    except Exception:
        return None
    if n != 1_000_000_000 or j.get("kernel") != kernel:
        return None
    return j["traffic_bytes_per_launch"]


def cpu_baseline(text, q_buf, q_off, sample_n, sample_q):
    """Reference libsais64_omp (oracle/_ref, compiled from the reference's own sources) on a
    bounded prefix of the same text, all host cores; falls back to the oracle port."""
    from oracle.oracle import Oracle, Ref
    t = np.ascontiguousarray(text[:sample_n])
    cores = usable_cpus()
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    os.environ.setdefault("OMP_DYNAMIC", "false")
    out = {}
    orc = Oracle()
    if Ref.available():
        ref = Ref()
        sa64 = np.zeros(t.size, dtype=np.int64)  # first-touched
        t0 = time.perf_counter()
        rc = ref.libsais64_into(t, sa64, threads)
        dt = time.perf_counter() - t0
        assert rc == 0
        out.update(kind="reference", value=t.size / dt, unit="chars/s", cores=threads,
                   sample=f"libsais64_omp(threads={threads}; {cores} cores visible, {os.cpu_count()} in the machine) on the first {t.size:,} chars of the same text, 1 run: {dt:.2f} s")
        sa = sa64.astype(np.uint32)
    else:
        t0 = time.perf_counter()
        sa = orc.sais(t).astype(np.uint32)
        dt = time.perf_counter() - t0
        out.update(kind="port", value=t.size / dt, unit="chars/s", cores=1,
                   sample=f"oracle SA-IS port on the first {t.size:,} chars, 1 run: {dt:.2f} s")
    # query baseline: oracle restatement of get_substring_positions, OpenMP over the batch
    nq = min(sample_q, q_off.size - 1)
    t0 = time.perf_counter()
    orc.query_batch(t, sa, 0xFFFFFFFF, (q_buf[:int(q_off[nq])], q_off[:nq + 1]), threads=threads)
    dq = time.perf_counter() - t0
    out["queries_per_s"] = nq / dq
    out["query_sample"] = f"{nq:,} of the same 16-byte patterns over the sample SA, {orc.threads_used} threads: {dq:.2f} s"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chars", "--n", dest="n", type=int, default=1_000_000_000)
    ap.add_argument("--queries", "--q", dest="q", type=int, default=1_000_000)
    ap.add_argument("--pattern-len", type=int, default=16)
    ap.add_argument("--cpu-sample", type=int, default=100_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exercise-dist", action="store_true",
                    help="run the multi-GPU code paths (RCCL init, result all-gather, index broadcast) even at world size 1")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from suffixarray_amd import _capi, synth
    from suffixarray_amd.distributed import broadcast_index

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or (args.exercise_dist and "RANK" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    N, Q, m = args.n, args.q, args.pattern_len
    text = synth.d1_uniform27(N)                     # same text on every rank
    q_buf, q_off = synth.query_batch(text, Q, m, seed=rank)  # each rank's own slice of the global batch

    idx = _capi.DeviceIndex(N, local_rank)
    idx.build(text)                                  # uploads the text into the index's HBM buffer
    text_dev = idx.text_dev
    pat_t = torch.from_numpy(np.concatenate([q_buf, np.zeros(64, np.uint8)])).to(dev)
    off_t = torch.from_numpy(q_off.view(np.int64)).to(dev)
    out_t = torch.zeros(2 * Q, dtype=torch.int32, device=dev)

    gathered = torch.empty(world * 2 * Q, dtype=torch.int32, device=dev) if use_dist else None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        idx.build_device(text_dev, N, 0)
        idx.query_batch_device(pat_t.data_ptr(), off_t.data_ptr(), Q, out_t.data_ptr())
        idx.sync()
        if use_dist:   # hits gathered on every rank (8 bytes per query over RCCL)
            dist.all_gather_into_tensor(gathered, out_t)

    for _ in range(args.warmup):
        step()
    barrier()
    build_ms, radix_ms, radix_launches, radix_bytes, query_ms = 0.0, 0.0, 0, 0, 0.0
    kind_ms, kind_bytes, kind_launches = [0.0] * 4, [0] * 4, [0] * 4
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = idx.build_stats()
        build_ms += st["total_ms"]
        radix_ms += st["radix_ms"]
        radix_launches += st["radix_passes"]
        radix_bytes += st["radix_bytes"]
        for k in range(4):
            kind_ms[k] += st["pass_ms"][k]; kind_bytes[k] += st["pass_bytes"][k]; kind_launches[k] += st["pass_launches"][k]
        query_ms += idx.query_stats()["kernel_ms"]
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt, build_ms, query_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, build_ms_max, query_ms_max = tmax.tolist()
    else:
        build_ms_max, query_ms_max = build_ms, query_ms
    last = idx.build_stats()

    # one-time replication cost (north_star): RCCL broadcast of text + SA from rank 0
    # The north-star replication path, outside the timed steps: rank 0's index (text + SA) reaches the
    # other GPUs by one RCCL broadcast per tensor, every rank adopts the copy (sa_hip_index_load_device)
    # and answers its slice of the batch from it; the ranges must equal those of its own build.
    bcast_ms, replica_ok = None, None
    if use_dist:
        if rank == 0:
            tx_t = torch.from_numpy(text).to(dev)
            sa_t = torch.from_numpy(idx.sa_u32().view(np.int32)).to(dev)
        else:
            tx_t = torch.empty(N, dtype=torch.uint8, device=dev)
            sa_t = torch.empty(N, dtype=torch.int32, device=dev)
        barrier()
        b0 = time.perf_counter()
        broadcast_index(tx_t, sa_t, src=0)
        barrier()
        bcast_ms = (time.perf_counter() - b0) * 1e3
        own = out_t.clone()
        rep = _capi.DeviceIndex(N, local_rank)
        rep.load_device(tx_t.data_ptr(), sa_t.data_ptr(), N, 0)
        rep.query_batch_device(pat_t.data_ptr(), off_t.data_ptr(), Q, out_t.data_ptr())
        rep.sync()
        ok_t = torch.tensor([1 if torch.equal(own, out_t) else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        replica_ok = bool(ok_t.item())
        rep.close()
        del sa_t, tx_t

    # correctness gate inside the bench: the SA of the last step is a suffix array (spot checks)
    gate = None
    if rank == 0:
        res = out_t.cpu().numpy().view(np.uint32).reshape(-1, 2)
        hits = ((res[:, 1].astype(np.int64) - res[:, 0].astype(np.int64) + 1) & 0xFFFFFFFF) > 0
        hits &= res[:, 0] != 0xFFFFFFFF
        probe = idx.sa_range(0, min(N, 1 << 16)).astype(np.int64)
        ok = True
        for a, b in zip(probe[:-1:97], probe[1::97]):
            ok &= bytes(text[a:a + 64]) <= bytes(text[b:b + 64])
        gate = {"sorted_probe_ok": bool(ok), "query_hit_rate": float(hits.mean())}

    if rank == 0:
        steps = args.steps
        chars_per_s = world * N * steps / (build_ms_max / 1e3)          # replicas: every rank builds N chars
        queries_per_s = world * Q * steps / (query_ms_max / 1e3)
        if last.get("text_top_pass"):
            PASS_KERNELS[1] = "text_top_pass_kernel<512>"
        # dominant kernel = the sort-pass kernel with the largest share of the timed region
        dom = max(range(4), key=lambda k: kind_ms[k])
        pass_ms = kind_ms[dom] / max(kind_launches[dom], 1)
        bytes_per_launch = kind_bytes[dom] / max(kind_launches[dom], 1)
        achieved = kind_bytes[dom] / (kind_ms[dom] / 1e3) if kind_ms[dom] > 0 else 0.0
        all_achieved = radix_bytes / (radix_ms / 1e3) if radix_ms > 0 else 0.0
        bq = 2 * int(np.ceil(np.log2(max(N, 2)))) * (4 + m)              # reference bytes per query
        q_achieved = Q * steps * bq / (query_ms / 1e3) if query_ms > 0 else 0.0
        line = {
            "metric": "sa_build_chars_per_s",
            "value": chars_per_s,
            "unit": "chars/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("u8 text / u32 narrow keys / u32 suffix indices" if last.get("narrow_k")
                      else "u8 text / u64 keys / u32 suffix indices"),
            "data": "synthetic",
            "config": {"workload": f"D1 uniform27 text N={N:,} (32-bit device build, libsais64-compatible 64-bit output by widening kernel) + {Q:,} batched {m}-byte queries per GPU",
                       "n_chars": N, "queries_per_gpu": Q, "pattern_len": m,
                       "parallelism": "replicas (build) + sharded query batch" if world > 1 else "single GPU"},
            "build_ms": build_ms_max / steps,
            "queries_per_s": queries_per_s,
            "query_ms": query_ms_max / steps,
            "broadcast_ms": bcast_ms,
            "replica_query_ok": replica_ok,
            "build_stats": {k: last[k] for k in ("sigma", "bits_per_symbol", "initial_chars", "rounds", "chunk_rounds",
                                                 "doubling_rounds", "final_depth", "radix_passes", "active_total", "narrow_k")},
            "roofline": {"bound": "hbm", "kernel": PASS_KERNELS[dom], "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK,
                         "traffic": pmc_traffic(N, PASS_KERNELS[dom]),
                         "traffic_note": "bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE on this workload, profiles/pmc_onesweep.json",
                         "launches": kind_launches[dom], "avg_launch_ms": pass_ms, "bytes_per_launch": bytes_per_launch},
            "sort_passes": {"achieved": all_achieved / 1e9, "unit": "GB/s", "frac": all_achieved / HBM_PEAK, "launches": radix_launches,
                            "by_kernel": {PASS_KERNELS[k]: {"launches": kind_launches[k], "avg_launch_ms": kind_ms[k] / kind_launches[k],
                                                            "bytes_per_launch": kind_bytes[k] / kind_launches[k],
                                                            "achieved": kind_bytes[k] / kind_ms[k] / 1e6}
                                          for k in range(4) if kind_launches[k]}},
            "roofline_query": {"bound": "hbm", "kernel": "query_kernel<true>" if last.get("narrow_k") else "query_kernel<false>", "achieved": q_achieved / 1e9, "peak": HBM_PEAK / 1e9,
                               "unit": "GB/s", "frac": q_achieved / HBM_PEAK, "bytes_per_query_model": bq,
                               "note": "bytes of the REFERENCE algorithm per query (SURVEY 8d: 2*ceil(log2 N)*(4+m)); the directory and the "
                                       "key array replace most of its probes, so the kernel moves far fewer real bytes and this figure can exceed the peak"},
            "gate": gate,
        }
        if last.get("narrow_k") and last.get("text_top_pass") and last.get("rounds") == 0:
            # the whole build with its own bytes / time: the sort passes as accounted above + per character
            # byte histogram 1, top-digit histogram 1, bucket histogram 4, flags pass 4 + 1, compaction 1
            other = 12.0
            total_bytes = radix_bytes + other * N * steps
            line["whole_build"] = {"bytes_per_char_model": total_bytes / (N * steps), "achieved": total_bytes / (build_ms / 1e3) / 1e9,
                                   "unit": "GB/s", "frac": total_bytes / (build_ms / 1e3) / HBM_PEAK}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(text, q_buf, q_off, min(args.cpu_sample, N), min(Q, 1_000_000))
        print(json.dumps(line))
    idx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
