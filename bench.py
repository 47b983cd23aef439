#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric):
SA build chars/s (N = 1e9) + batched queries/s (Q = 1e6) on one MI355X, % of HBM roofline.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--chars CHARS] [--queries QUERIES]

N = 1 (BASELINE config 3: "1 GB text (libsais64 path), 64-bit SA build + 1M batched 16-byte queries"):
a "step" = one device build of the suffix array of the N-char synthetic text D1 (uniform27; text
resident in HBM before the timed region) + the widening pass that leaves the result in libsais64 layout
(int64[N], device resident) + one batch of Q 16-byte queries (50 % text windows, 50 % random; patterns
resident in HBM).  `value` = N / (build + widen time): every byte config 3 names is inside it.

N > 1 (BASELINE config 4: "1 GB text, 10M-query batch sharded across 8 GPUs, SA replicated via RCCL, hits
gathered"; launched by torch.distributed.run, one rank per GPU): construction stays single-GPU -- rank 0 builds
ONCE, its query structures (text, SA, key array, directory) reach the other GPUs by one RCCL broadcast per buffer
straight into buffers the replicas have reserved (timed, `replicate_ms`; nothing is rebuilt there); a "step" = ONE global
batch of --queries-global patterns (the same on every rank) split into contiguous slices, searched chunk by chunk with no
data-path collective while the 8-byte ranges of the previous chunk are gathered (--dist-mode all_gather | gather_to_root).
`value` = global queries/s (strong scaling); the build's chars/s is reported un-multiplied beside it.
Prints ONE JSON line on rank 0.
"""
import os

os.environ.setdefault("OMP_DYNAMIC", "false")   # libsais turns its parallel regions off when omp_get_dynamic() (libsais.c:744)

import argparse
import json
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


def committed_pmc(name, n, kernel):
    """HBM bytes from a committed rocprofv3 --pmc run (profiles/<name>; PMC passes cannot run inside this
    process).  Only reported for the workload and the kernel it was measured on."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None
    if n != j.get("n_chars", 1_000_000_000) or j.get("kernel") != kernel:
        return None
    return j


def cpu_baseline(text, q_buf, q_off, gpu_sa64, one_thread_n, one_run=False):
    """The reference's libsais64_omp (oracle/_ref, compiled from the reference's own sources) on the SAME text at
    full size, all usable host cores, best of two runs; the 1-thread libsais figure (the configuration the
    reference's Makefile builds, Makefile:2-5) on a prefix; the oracle's OpenMP restatement of
    get_substring_positions over the same batch.  Also returns whether the GPU's int64 suffix array equals
    the reference's output (bit-exact at full size).  Falls back to the oracle port where oracle/_ref is absent."""
    from oracle.oracle import Oracle, Ref
    cores = usable_cpus()
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    out = {}
    orc = Oracle()
    n = text.size
    equal = None
    if Ref.available():
        ref = Ref()
        sa64 = np.zeros(n, dtype=np.int64)   # first-touched
        runs = []
        for _ in range(2):
            t0 = time.perf_counter()
            rc = ref.libsais64_into(text, sa64, threads)
            runs.append(time.perf_counter() - t0)
            assert rc == 0
            if runs[-1] > 40.0 or one_run:
                break
        dt = min(runs)
        out.update(kind="reference", value=n / dt, unit="chars/s", cores=threads,
                   sample=f"libsais64_omp(threads={threads}, OMP_DYNAMIC=false; {cores} cores usable, {os.cpu_count()} in the machine) "
                          f"on the whole text, n = {n:,}, runs: {', '.join('%.2f s' % r for r in runs)} (best)")
        if gpu_sa64 is not None:
            equal = bool(np.array_equal(gpu_sa64, sa64))
        sa = sa64.astype(np.uint32)
        del sa64
        m1 = min(one_thread_n, n)
        t1 = np.ascontiguousarray(text[:m1])
        s1 = np.zeros(m1, dtype=np.int32)
        t0 = time.perf_counter()
        rc = ref.lib.libsais(t1.ctypes.data, s1.ctypes.data, m1, 0, None)
        d1 = time.perf_counter() - t0
        assert rc == 0
        out["one_thread"] = {"value": m1 / d1, "unit": "chars/s", "cores": 1,
                             "sample": f"libsais (1 thread, the reference Makefile's configuration) on the first {m1:,} chars: {d1:.2f} s"}
        del s1
    else:
        m1 = min(one_thread_n, n)
        t1 = np.ascontiguousarray(text[:m1])
        t0 = time.perf_counter()
        sa = orc.sais(t1).astype(np.uint32)
        dt = time.perf_counter() - t0
        out.update(kind="port", value=m1 / dt, unit="chars/s", cores=1,
                   sample=f"oracle SA-IS port on the first {m1:,} chars, 1 run: {dt:.2f} s")
        text = t1
    # query baseline: oracle restatement of get_substring_positions, OpenMP over the batch
    nq = q_off.size - 1
    t0 = time.perf_counter()
    orc.query_batch(text, sa, 0xFFFFFFFF, (q_buf, q_off), threads=threads)
    dq = time.perf_counter() - t0
    out["queries_per_s"] = nq / dq
    out["query_sample"] = f"{nq:,} of the same 16-byte patterns over the SA of the {text.size:,}-char text, {orc.threads_used} threads: {dq:.2f} s"
    return out, equal


def d2_words_1e9(_capi, synth, torch, dev, device, n=1_000_000_000, with_cpu=True):
    """SURVEY 8(d)'s realistic text at the HEADLINE size: D2 words, N = 1e9, the 64-bit build (u32 device build + int64[N] in
    libsais64 layout, device resident) -- the input on which the pipeline needs its group finisher and refinement rounds (D1
    needs none).  Verified on the device and, with the CPU baseline, compared with the reference's libsais64_omp output."""
    t0 = time.perf_counter()
    text = synth.d2_words_parts(n)
    gen_s = time.perf_counter() - t0
    sa64_t = torch.empty(n, dtype=torch.int64, device=dev)
    with _capi.DeviceIndex(n, device) as idx:
        idx.build(text)
        tdev = idx.text_dev
        ms = []
        for _ in range(3):
            idx.build_device64(tdev, n, sa64_t.data_ptr(), 0)
            ms.append(idx.build_stats()["total_ms"])
        st = idx.build_stats()
        out = {"n_chars": n, "text": "synth.d2_words_parts: Zipf(1.0) over one 50 000-word vocabulary, 16 independently drawn parts", "text_gen_s": gen_s,
               "build_ms": min(ms), "chars_per_s": n / (min(ms) / 1e3), "verify_violations": idx.verify(),
               "output": "u32 suffix array + int64[N] libsais64 layout, both device resident",
               **{k: st[k] for k in ("initial_chars", "rounds", "chunk_rounds", "doubling_rounds", "active_total", "finisher_runs", "finisher_records",
                                     "finisher_resolved", "narrow48", "widen_fused", "final_depth")}}
        sa32 = idx.sa_u32()
    gpu64 = sa64_t.cpu().numpy()
    del sa64_t
    out["sa64_equals_sa32"] = bool(np.array_equal(gpu64, sa32))
    del sa32
    if with_cpu:
        from oracle.oracle import Ref
        if Ref.available():
            cores = usable_cpus()
            threads = int(os.environ.get("OMP_NUM_THREADS", cores))
            ref64 = np.zeros(n, dtype=np.int64)
            t0 = time.perf_counter()
            rc = Ref().libsais64_into(text, ref64, threads)
            dt = time.perf_counter() - t0
            assert rc == 0
            out["sa64_equals_reference_libsais64"] = bool(np.array_equal(gpu64, ref64))
            out["cpu_reference"] = {"kind": "reference", "value": n / dt, "unit": "chars/s", "cores": threads,
                                    "sample": f"libsais64_omp(threads={threads}) on the whole D2 text, one run: {dt:.2f} s"}
            del ref64
    out["ok"] = bool(out["verify_violations"] == 0 and out["sa64_equals_sa32"] and out.get("sa64_equals_reference_libsais64", True))
    return out


def secondary_builds(_capi, synth, device):
    """Non-headline inputs at N = 1e8, driver-visible: config 2 (D1, 32-bit SA) and D2 words (refinement rounds).
    Device build time (HIP events), best of 3 after one warm-up; verified on the device."""
    out = {}
    n = 100_000_000
    for name, gen in (("config2_d1_1e8", lambda: synth.d1_uniform27(n)), ("d2_words_1e8", lambda: synth.d2_words(n))):
        text = gen()
        with _capi.DeviceIndex(n, device) as idx:
            idx.build(text)
            tdev = idx.text_dev
            ms = []
            for _ in range(3):
                idx.build_device(tdev, n, 0)
                ms.append(idx.build_stats()["total_ms"])
            st = idx.build_stats()
            out[name] = {"n_chars": n, "build_ms": min(ms), "chars_per_s": n / (min(ms) / 1e3), "verify_violations": idx.verify(),
                         "initial_chars": st["initial_chars"], "rounds": st["rounds"], "active_total": st["active_total"]}
        del text
    out["config5_csv_50M_rows"] = config5_csv(_capi)
    return out


def config5_csv(_capi, rows=50_000_000, budget_s=12.0):
    """BASELINE config 5 with the reference's own measurement protocol (tests/test.py:99-141): build from the CSV column,
    sample names from the column (upper-cased, test.py:103-106), time each query_records with perf_counter (122-127),
    report build seconds, mean and median microseconds and the mean number of results (131-136).  10 000 samples as
    in the reference, cut short when `budget_s` of query time is used up (the count is reported)."""
    import tempfile
    from suffixarray_amd import SuffixArray
    tmp = tempfile.mkdtemp(prefix="sa_hip_c5_")
    path = os.path.join(tmp, "companies.csv")
    try:
        _capi.synth_csv(path, rows, 1)
        t0 = time.perf_counter()
        s = SuffixArray(csv_file=path, search_column="company_name", max_suffix_length=32)
        t_index = time.perf_counter() - t0
        idx = s._index
        st = idx.build_stats()
        n = idx.n
        idx.build_device(idx.text_dev, n, 32)   # second build: allocations warm
        st2 = idx.build_stats()
        violations = idx.verify()
        rng = np.random.default_rng(0)
        # sample names straight from the file's rows (the column of `id,company_name,country`)
        size = os.path.getsize(path)
        sample = []
        with open(path, "rb") as f:
            for o in np.sort(rng.integers(0, size - 4096, 10_000)):
                f.seek(int(o))
                lines = f.read(4096).split(b"\n")
                if len(lines) >= 3:
                    sample.append(next(_csv_reader([lines[1].decode()]))[1].upper())
        def protocol():
            lat, nres = [], []
            t_all = time.perf_counter()
            for q in sample:
                t0 = time.perf_counter()
                r = s.query_records(q)
                lat.append((time.perf_counter() - t0) * 1e6)
                nres.append(len(r))
                if time.perf_counter() - t_all > budget_s:
                    break
            return np.array(lat), nres
        # the class gives its indexes their second-level keys at construction (round 4: sa_hip_index_deep_keys); the timed rebuild
        # above dropped them: back to the state SuffixArray(csv_file=...) leaves, then the same 10 000 queries without them
        idx.deep_keys(2)
        lat, nres = protocol()
        idx.deep_keys(0)
        lat_plain, nres_plain = protocol()
        idx.deep_keys(2)
        out = {"rows": rows, "file_bytes": size, "n_chars": n, "max_suffix_length": 32,
               "index_seconds_end_to_end": t_index, "device_build_ms_first": st["total_ms"], "device_build_ms": st2["total_ms"],
               "chars_per_s": n / (st2["total_ms"] / 1e3), "verify_violations": violations, "rounds": st2["rounds"],
               "finisher_resolved": st2["finisher_resolved"],
               "query_records": {"samples": int(lat.size), "mean_us": float(lat.mean()), "median_us": float(np.median(lat)),
                                 "mean_results": float(np.mean(nres)), "k": 1000,
                                 "protocol": "tests/test.py:99-141 (sampled names, upper-cased, perf_counter per query)",
                                 "without_deep_keys": {"mean_us": float(lat_plain.mean()), "median_us": float(np.median(lat_plain)),
                                                       "same_result_counts": bool(nres_plain == nres)}}}
        # the column itself (one name per '\n'-terminated row) feeds the two measurements below
        col = idx.text()
        out["names_batch_1e6"] = names_batch(idx, col, rng)
        # a batch large enough for the clustering of round 4 (from 2^20 patterns on: answered in the order of the patterns' first characters)
        out["names_batch_8e6"] = names_batch(idx, col, rng, q=8_000_000, rows_k=0)
        s.close()
        del s, idx
        out["documents_5M"] = documents_protocol(col, rng)
        return out
    finally:
        try:
            os.remove(path)
            os.rmdir(tmp)
        except OSError:
            pass


def names_batch(idx, col, rng, q=1_000_000, rows_k=16):
    """The reference's real query load -- names that occur (tests/test.py:103-127) -- as ONE batch: q rows sampled from the
    column, searched by one launch (sa_hip_query_batch: host patterns in, ranges out); kernel time from the library's events."""
    ends = np.flatnonzero(col == 10)
    pick = np.sort(rng.integers(1, ends.size, q))
    a, b = ends[pick - 1] + 1, ends[pick]
    keep = b > a
    a, b = a[keep], b[keep]
    lens = (b - a).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    src = np.repeat(a - off[:-1].astype(np.int64), lens.astype(np.int64)) + np.arange(int(off[-1]), dtype=np.int64)
    buf = col[src]

    def run(reps):
        ms, res = [], None
        for _ in range(reps):
            res = idx.query_batch((buf, off))
            ms.append(idx.query_stats()["kernel_ms"])
        return min(ms[1:]), res

    # without the second-level keys (round 3's search: text comparisons inside a key group), then with them (round 4)
    idx.deep_keys(0)
    plain_ms, plain = run(3)
    t0 = time.perf_counter()
    has = idx.deep_keys(2)
    k2_build_ms = (time.perf_counter() - t0) * 1e3
    best, res = run(4)
    cnt = ((res["second"].astype(np.int64) - res["first"].astype(np.int64) + 1) & 0xFFFFFFFF)
    cnt[res["first"] == 0xFFFFFFFF] = 0
    rows_out = None
    if rows_k:
        # ... and the same batch all the way to row ids (what query_records_batch needs: sa_hip_index_query_rows_batch, host patterns
        # in, up to rows_k distinct row ids per name out; ranges of <= 4 hits by one lane each, longer ones by a workgroup each)
        keep_arrays = (np.empty((a.size, rows_k), dtype=np.uint64), np.zeros(a.size, dtype=np.uint32),
                       np.zeros(a.size, dtype=[("first", "<u4"), ("second", "<u4")]))
        call_ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            (rw, rc_), rr = idx.query_rows_batch_raw((buf, off), rows_k, out=keep_arrays)
            call_ms.append((time.perf_counter() - t0) * 1e3)
        rows_out = {"k": rows_k, "call_ms": min(call_ms[1:]), "queries_per_s": a.size / (min(call_ms[1:]) / 1e3),
                    "mean_rows": float(rc_.mean()), "ranges_of_at_most_4_hits": float((cnt <= 4).mean()),
                    "same_ranges_as_the_search": bool(np.array_equal(rr["first"], res["first"]) and np.array_equal(rr["second"], res["second"])),
                    "counts_consistent": bool(np.array_equal(rc_ > 0, cnt > 0) and int(rc_.max()) <= rows_k),
                    "note": "host clock around the call: patterns up, search, rows kernels, counts + ranges + row ids down into the caller's arrays"}
    return {"queries": int(a.size), "rows_batch": rows_out, "mean_pattern_len": float(lens.mean()), "kernel_ms": best, "queries_per_s": a.size / (best / 1e3),
            "hit_rate": float((cnt > 0).mean()), "mean_hits": float(cnt.mean()), "median_hits": float(np.median(cnt)),
            "deep_keys": {"built": bool(has), "build_ms_host_clock": k2_build_ms, "same_ranges": bool(np.array_equal(res, plain)),
                          "without": {"kernel_ms": plain_ms, "queries_per_s": a.size / (plain_ms / 1e3)}},
            "note": "every pattern is a whole name of the column (all hit, longer than the 11-character key): with the second-level keys "
                    "(sa_hip_index_deep_keys: built once per index, 8 n bytes) the bounds inside a key group come from a search over "
                    "8-byte keys, without them from text comparisons; kernel time, patterns resident"}


def documents_protocol(col, rng, docs_n=5_000_000, budget_s=10.0):
    """tests/test.py:59-96, the documents constructor: the (upper-cased) company names as a document LIST, construction
    seconds, 10 000 sampled documents as queries with perf_counter around each query_records, mean / median microseconds
    and the mean number of results."""
    from suffixarray_amd import SuffixArray
    ends = np.flatnonzero(col == 10)
    cut = int(ends[min(docs_n, ends.size) - 1]) + 1
    docs = col[:cut].tobytes().decode("latin-1").upper().split("\n")[:-1]
    del ends
    t0 = time.perf_counter()
    s = SuffixArray(documents=docs, max_suffix_length=32)
    t_index = time.perf_counter() - t0
    st = s._index.build_stats()
    sample = [docs[i] for i in rng.integers(0, len(docs), 10_000)]
    lat, nres = [], []
    t_all = time.perf_counter()
    for q in sample:
        t0 = time.perf_counter()
        r = s.query_records(q)
        lat.append((time.perf_counter() - t0) * 1e6)
        nres.append(len(r))
        if time.perf_counter() - t_all > budget_s:
            break
    lat = np.array(lat)
    out = {"documents": len(docs), "n_chars": int(st["n"]), "max_suffix_length": 32, "construction_seconds": t_index,
           "device_build_ms": st["total_ms"],
           "query_records": {"samples": int(lat.size), "mean_us": float(lat.mean()), "median_us": float(np.median(lat)),
                             "mean_results": float(np.mean(nres)), "k": 1000,
                             "protocol": "tests/test.py:59-96 (document list, sampled documents upper-cased, perf_counter per query)"}}
    s.close()
    return out


def _csv_reader(lines):
    import csv
    return csv.reader(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chars", "--n", dest="n", type=int, default=1_000_000_000)
    ap.add_argument("--queries", "--q", dest="q", type=int, default=1_000_000, help="batch size of the 1-GPU step")
    ap.add_argument("--queries-global", type=int, default=10_000_000, help="global batch of the N > 1 step (config 4)")
    ap.add_argument("--pattern-len", type=int, default=16)
    ap.add_argument("--cpu-one-thread-chars", type=int, default=100_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-beyond-u32", action="store_true", help="skip secondary.beyond_u32_4p4e9 (a 4.4e9-character text: 64-bit suffix indices, csrc/big_build.hpp)")
    ap.add_argument("--no-words-1e9", action="store_true", help="skip secondary.d2_words_1e9 (D2 words at N = 1e9, 64-bit build + the reference's libsais64_omp)")
    ap.add_argument("--offsets-api", action="store_true", help="query through sa_hip_query_batch_device (offsets array) instead of the fixed-length entry")
    ap.add_argument("--separate-widen", action="store_true", help="int64 output by a widening pass after the build (A/B against the fused form)")
    ap.add_argument("--dist-chunks", type=int, default=4, help="pieces a rank's slice is searched and gathered in (pipelined); at most this many, none under 1e6 patterns")
    ap.add_argument("--dist-min-chunk", type=int, default=1_000_000, help="patterns a pipelined piece carries at least")
    ap.add_argument("--dist-mode", choices=("all_gather", "gather_to_root", "sharded_rows"), default="all_gather",
                    help="all_gather / gather_to_root: config 4 as written (the 8-byte ranges of the whole batch land on every rank / on rank 0); "
                         "sharded_rows: the ranges stay on the rank that found them, every rank materialises the row ids of ITS slice "
                         "(sa_hip_index_query_rows_batch), only per-rank counts are reduced -- the mode whose rate can scale with the ranks")
    ap.add_argument("--rows-k", type=int, default=16, help="sharded_rows: distinct rows returned per query at most")
    ap.add_argument("--dump", default=None, help="config 4: write the gathered ranges (+ the SA for N <= 1e8) to this .npz (tests)")
    ap.add_argument("--exercise-dist", action="store_true",
                    help="run the multi-GPU path (RCCL init, index broadcast, sharded batch, all-gather) at world size 1")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from suffixarray_amd import _capi, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.exercise_dist:
        if "RANK" not in os.environ:   # --exercise-dist started by hand
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29511"))
        dist.init_process_group("nccl", device_id=dev)
        line = run_sharded(args, torch, dist, _capi, synth, rank, local_rank, world, dev)
        if rank == 0:
            print(json.dumps(line))
        dist.destroy_process_group()
        return
    print(json.dumps(run_single(args, torch, _capi, synth, dev, local_rank)))


def run_single(args, torch, _capi, synth, dev, device):
    N, Q, m = args.n, args.q, args.pattern_len
    text = synth.d1_uniform27(N)
    q_buf, q_off = synth.query_batch(text, Q, m, seed=0)

    idx = _capi.DeviceIndex(N, device)
    idx.build(text)                                  # uploads the text into the index's HBM buffer
    text_dev = idx.text_dev
    pat_t = torch.from_numpy(np.concatenate([q_buf, np.zeros(64, np.uint8)])).to(dev)
    off_t = torch.from_numpy(q_off.view(np.int64)).to(dev)
    out_t = torch.zeros(2 * Q, dtype=torch.int32, device=dev)
    sa64_t = torch.empty(N, dtype=torch.int64, device=dev)    # libsais64 layout, device resident

    def step():
        # the 64-bit build: u32 suffix array (the index's own, searched by the queries) AND int64[N] in libsais64 layout;
        # --separate-widen: the int64 copy by a widening pass of its own instead of out of the sort's last pass
        if args.separate_widen:
            idx.build_device(text_dev, N, 0)
            idx.widen_device(sa64_t.data_ptr())
        else:
            idx.build_device64(text_dev, N, sa64_t.data_ptr(), 0)
        # every pattern has m bytes: the fixed-length entry point (no offsets array to read); --offsets-api: the general one
        if args.offsets_api:
            idx.query_batch_device(pat_t.data_ptr(), off_t.data_ptr(), Q, out_t.data_ptr())
        else:
            idx.query_batch_device_fixed(pat_t.data_ptr(), m, Q, out_t.data_ptr())
        idx.sync()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    build_ms, widen_ms, radix_ms, radix_launches, radix_bytes, query_ms = 0.0, 0.0, 0.0, 0, 0, 0.0
    kind_ms, kind_bytes, kind_launches = [0.0] * 4, [0] * 4, [0] * 4
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = idx.build_stats()
        build_ms += st["total_ms"]
        widen_ms += st["widen_ms"]
        radix_ms += st["radix_ms"]
        radix_launches += st["radix_passes"]
        radix_bytes += st["radix_bytes"]
        for k in range(4):
            kind_ms[k] += st["pass_ms"][k]; kind_bytes[k] += st["pass_bytes"][k]; kind_launches[k] += st["pass_launches"][k]
        query_ms += idx.query_stats()["kernel_ms"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last = idx.build_stats()
    steps = args.steps

    # correctness gates, outside the timed region: on-device sufcheck of the last step's SA (the SA is unique:
    # verified == bit-exact), and -- with the CPU baseline -- equality with the reference's libsais64 output
    res = out_t.cpu().numpy().view(np.uint32).reshape(-1, 2)
    hits = ((res[:, 1].astype(np.int64) - res[:, 0].astype(np.int64) + 1) & 0xFFFFFFFF) > 0
    hits &= res[:, 0] != 0xFFFFFFFF
    violations = idx.verify()
    gate = {"verify_violations": violations, "query_hit_rate": float(hits.mean())}
    populations = query_populations(idx, synth, torch, dev, text, Q, m)

    total_ms = build_ms + widen_ms
    chars_per_s = N * steps / (total_ms / 1e3)
    queries_per_s = Q * steps / (query_ms / 1e3)
    if last.get("text_top_pass"):
        PASS_KERNELS[1] = "text_top_pass_kernel<512>"
    if last.get("split_plan"):   # the three-pass plan (radix_split.hpp): kinds 2 / 3 are its two launches
        PASS_KERNELS[2] = "seg_split_kernel<512, 24>"
        PASS_KERNELS[3] = "local_finish_kernel"
    # dominant kernel = the sort-pass kernel with the largest share of the timed region
    dom = max(range(4), key=lambda k: kind_ms[k])
    pass_ms = kind_ms[dom] / max(kind_launches[dom], 1)
    bytes_per_launch = kind_bytes[dom] / max(kind_launches[dom], 1)
    achieved = kind_bytes[dom] / (kind_ms[dom] / 1e3) if kind_ms[dom] > 0 else 0.0
    all_achieved = radix_bytes / (radix_ms / 1e3) if radix_ms > 0 else 0.0
    bq = 2 * int(np.ceil(np.log2(max(N, 2)))) * (4 + m)              # reference bytes per query
    q_model = Q * steps * bq / (query_ms / 1e3) if query_ms > 0 else 0.0
    pmc_sort = committed_pmc("pmc_onesweep.json", N, PASS_KERNELS[dom])
    qk = "query_kernel<true, 2>" if last.get("narrow_k") else "query_kernel<false, 2>"   # 2 = 32-byte key windows (the product path)
    pmc_q = committed_pmc("pmc_query.json", N, qk)
    rq = {"bound": "hbm", "kernel": qk, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
          "reference_model": {"bytes_per_query": bq, "achieved": q_model / 1e9,
                              "note": "bytes of the REFERENCE algorithm per query (SURVEY 8d: 2*ceil(log2 N)*(4+m)); the directory and the key "
                                      "array replace most of its probes, so this is not what the kernel moves"}}
    if pmc_q and pmc_q.get("queries") == Q and not args.offsets_api:
        # measured HBM bytes of one batch (rocprofv3 --pmc, profiles/pmc_query.json) over THIS run's kernel time
        tq = pmc_q["traffic_bytes_per_launch"]
        rq.update(traffic=tq, bytes_per_query=tq / Q, achieved=tq / (query_ms / steps / 1e3) / 1e9,
                  frac=tq / (query_ms / steps / 1e3) / HBM_PEAK,
                  traffic_note="FETCH_SIZE (x1: random 64-byte requests are tallied exactly) + WRITE_SIZE of one 1M-query launch, profiles/pmc_query.json")
        # the batch is bound by the NUMBER of random 64-byte requests, not by bytes: requests per second against the rate
        # the memory system sustains for that access pattern (tools/gatherbench, profiles/r02_a_pmc_calibration.md)
        if pmc_q.get("read_requests_per_query") and pmc_q.get("random_read_requests_per_s_ceiling"):
            rps = pmc_q["read_requests_per_query"] * Q / (query_ms / steps / 1e3)
            rq["random_requests"] = {"per_query": pmc_q["read_requests_per_query"], "per_s": rps,
                                     "ceiling_per_s": pmc_q["random_read_requests_per_s_ceiling"],
                                     "frac_of_ceiling": rps / pmc_q["random_read_requests_per_s_ceiling"]}
    else:
        rq.update(traffic=None, achieved=None, frac=None)
    line = {
        "metric": "sa_build_chars_per_s",
        "value": chars_per_s,
        "unit": "chars/s",
        "n_gpus": 1,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("u8 text / u32 narrow keys / u32 suffix indices / i64 output" if last.get("narrow_k")
                  else "u8 text / u64 keys / u32 suffix indices / i64 output"),
        "data": "synthetic",
        "config": {"workload": f"config 3: D1 uniform27 text N={N:,} (libsais64 path), 64-bit SA build (u32 device build, int64[N] libsais64 layout "
                               f"{'written by the last sort pass' if last.get('widen_fused') else 'by a widening pass'}, device resident) "
                               f"+ {Q:,} batched {m}-byte queries, 1 GPU",
                   "n_chars": N, "queries": Q, "pattern_len": m, "parallelism": "single GPU"},
        "build_ms": total_ms / steps,
        "build_ms_u32": build_ms / steps,
        "widen_ms": widen_ms / steps,
        "chars_per_s_u32": N * steps / (build_ms / 1e3),
        "queries_per_s": queries_per_s,
        "query_ms": query_ms / steps,
        "build_stats": {k: last[k] for k in ("sigma", "bits_per_symbol", "initial_chars", "rounds", "chunk_rounds",
                                             "doubling_rounds", "final_depth", "radix_passes", "active_total", "narrow_k",
                                             "split_plan", "split_max")},
        "roofline": {"bound": "hbm", "kernel": PASS_KERNELS[dom], "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK,
                     # PMC passes cannot run inside this process: `traffic` is REPLAYED from the committed counter run of the same
                     # kernel on the same workload (profiles/pmc_onesweep.json), not measured in this run
                     "traffic": pmc_sort["traffic_bytes_per_launch"] if pmc_sort else None,
                     "traffic_source": "committed_pmc" if pmc_sort else None,
                     "traffic_committed_pmc": pmc_sort["traffic_bytes_per_launch"] if pmc_sort else None,
                     "traffic_note": "bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE on this workload, REPLAYED from the committed run "
                                     "profiles/pmc_onesweep.json (keyed on kernel name and N), not collected in this process",
                     "launches": kind_launches[dom], "avg_launch_ms": pass_ms, "bytes_per_launch": bytes_per_launch},
        "sort_passes": {"achieved": all_achieved / 1e9, "unit": "GB/s", "frac": all_achieved / HBM_PEAK, "launches": radix_launches,
                        "by_kernel": {PASS_KERNELS[k]: {"launches": kind_launches[k], "avg_launch_ms": kind_ms[k] / kind_launches[k],
                                                        "bytes_per_launch": kind_bytes[k] / kind_launches[k],
                                                        "achieved": kind_bytes[k] / kind_ms[k] / 1e6}
                                      for k in range(4) if kind_launches[k]}},
        "widen": ({"fused": True, "note": ("int64 stores in local_finish_kernel" if last.get("split_plan") else "int64 stores in seg_onesweep_kernel<512, 24, true, true>") + " (8 more bytes per record) + a patch of the refined slots"}
                  if last.get("widen_fused") else
                  {"fused": False, "bytes_per_launch": 12.0 * N, "avg_launch_ms": widen_ms / steps,
                   "achieved": 12.0 * N * steps / (widen_ms / 1e3) / 1e9 if widen_ms > 0 else None, "unit": "GB/s"}),
        "roofline_query": rq,
        "query_populations": populations,
        "gate": gate,
    }
    if last.get("narrow_k") and last.get("text_top_pass") and last.get("rounds") == 0:
        # the whole build with its own bytes / time: the sort passes as accounted above + per character
        # byte histogram 1, top-digit histogram 1, bucket histogram 4, flags pass 4 + 1, compaction 1; the int64
        # output: 8 bytes per character inside the last sort pass (already in its pass bytes) or 12 as a pass of its own
        other = 12.0 + (0.0 if last.get("widen_fused") else 12.0)
        if last.get("split_plan") and last.get("lite_flags") == 2:
            # three-pass plan with the flags work inside the local pass: byte histogram 1, top-digit histogram 1, split histogram 4
            # (no flags pass, no compaction: the directory slice and the tied slots leave with the local pass)
            other = 6.0 + (0.0 if last.get("widen_fused") else 12.0)
        total_bytes = radix_bytes + other * N * steps
        line["whole_build"] = {"bytes_per_char_model": total_bytes / (N * steps), "achieved": total_bytes / (total_ms / 1e3) / 1e9,
                               "unit": "GB/s", "frac": total_bytes / (total_ms / 1e3) / HBM_PEAK}
    gpu_sa64 = sa64_t.cpu().numpy() if not (args.no_cpu_baseline and args.no_secondary) else None
    del sa64_t
    if not args.no_cpu_baseline:
        base, equal = cpu_baseline(text, q_buf, q_off, gpu_sa64, args.cpu_one_thread_chars)
        line["cpu_baseline"] = base
        if equal is not None:
            gate["sa64_equals_reference_libsais64"] = equal
    gate["ok"] = bool(violations == 0 and gate.get("sa64_equals_reference_libsais64", True))
    idx.close()
    if not args.no_secondary:
        torch.cuda.empty_cache()
        dropin = dropin_calls(_capi, torch, dev, text, gpu_sa64)
        del gpu_sa64, text
        line["secondary"] = secondary_builds(_capi, synth, device)
        line["secondary"]["dropin"] = dropin
        if N >= 1_000_000_000 and not args.no_words_1e9:
            torch.cuda.empty_cache()
            line["secondary"]["d2_words_1e9"] = d2_words_1e9(_capi, synth, torch, dev, device, with_cpu=not args.no_cpu_baseline)
            gate["d2_words_1e9_ok"] = line["secondary"]["d2_words_1e9"]["ok"]
            gate["ok"] = bool(gate["ok"] and gate["d2_words_1e9_ok"])
        if N >= 1_000_000_000 and not args.no_beyond_u32:
            _capi.release_workspace()
            torch.cuda.empty_cache()
            big = beyond_u32(_capi, synth, torch, dev, device)
            if big is not None:
                line["secondary"]["beyond_u32_4p4e9"] = big
                gate["beyond_u32_ok"] = big["ok"]
                gate["ok"] = bool(gate["ok"] and big["ok"])
    return line


def beyond_u32(_capi, synth, torch, dev, device, n=4_400_000_000):
    """A single text of more than 2^32 - 2 bytes (round 4: csrc/big_build.hpp, what sa_hip_libsais64 runs there -- the counterpart
    of libsais64.c:6684 -> libsais64_main): D1 at n = 4.4e9 built with 64-bit suffix indices on device buffers, checked by the
    64-bit sufcheck on the device.  None when the GPU has less than 46 bytes of free memory per character."""
    free, _ = torch.cuda.mem_get_info(device)
    if free < 46 * n:
        return None
    t0 = time.perf_counter()
    text = synth.d1_uniform27(n)
    gen_s = time.perf_counter() - t0
    text_t = torch.from_numpy(text).to(dev)
    sa_t = torch.empty(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = _capi.libsais64_device(text_t.data_ptr(), sa_t.data_ptr(), n, device)   # (one build: the call allocates and frees 140 GB around it)
    call_s = time.perf_counter() - t0
    ms = [st["total_ms"]]
    bad = _capi.sufcheck64_device(text_t.data_ptr(), sa_t.data_ptr(), n, device)
    beyond = int((sa_t > 0xFFFFFFFF).sum().item())
    del sa_t, text_t
    torch.cuda.empty_cache()
    return {"n_chars": n, "text": "D1 uniform27", "text_gen_s": gen_s, "build_ms": min(ms), "call_seconds_with_allocations": call_s, "chars_per_s": n / (min(ms) / 1e3),
            "sufcheck64_violations": bad, "entries_beyond_2_32": beyond, "output": "int64[N] libsais64 layout, device resident (64-bit suffix indices throughout)",
            **{k: st[k] for k in ("sigma", "bits_per_symbol", "initial_chars", "sort_passes", "rounds", "tied_after_sort")},
            "ok": bool(bad == 0 and beyond == n - (1 << 32))}


def query_populations(idx, synth, torch, dev, text, Q, m):
    """The two sub-populations of the D1 batch on their own ("keep both sub-populations and report hit rate"): a
    batch of Q patterns that all HIT (text windows without a newline: directory + key window + SA + text for a pattern longer
    than the key) and one of Q that all MISS (random strings: most die in the key array).  Kernel time, best of 5."""
    big_buf, big_off = synth.query_batch(text, 8 * Q, m, seed=7)
    pats = big_buf.reshape(-1, m)
    res = idx.query_batch((big_buf, big_off))
    hit = (((res["second"].astype(np.int64) - res["first"].astype(np.int64) + 1) & 0xFFFFFFFF) > 0) & (res["first"] != 0xFFFFFFFF)
    out = {}
    for name, sel in (("hits_only", np.flatnonzero(hit)[:Q]), ("misses_only", np.flatnonzero(~hit)[:Q])):
        if sel.size == 0:
            continue
        p_t = torch.from_numpy(np.concatenate([np.ascontiguousarray(pats[sel]).reshape(-1), np.zeros(64, np.uint8)])).to(dev)
        o_t = torch.empty(2 * sel.size, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ms = []
        for _ in range(6):
            idx.query_batch_device_fixed(p_t.data_ptr(), m, int(sel.size), o_t.data_ptr())
            idx.sync()
            ms.append(idx.query_stats()["kernel_ms"])
        r = o_t.cpu().numpy().view(np.uint32).reshape(-1, 2)
        h = (((r[:, 1].astype(np.int64) - r[:, 0].astype(np.int64) + 1) & 0xFFFFFFFF) > 0) & (r[:, 0] != 0xFFFFFFFF)
        best = min(ms[1:])
        out[name] = {"queries": int(sel.size), "kernel_ms": best, "queries_per_s": sel.size / (best / 1e3), "hit_rate": float(h.mean())}
        del p_t, o_t
    return out


def dropin_calls(_capi, torch, dev, text, gpu_sa64):
    """The libsais-call-compatible entry points as a caller of the reference would use them (main.c:70-76): host
    pointers in, host suffix array out -- wall time of sa_hip_libsais64(T, SA, N) at the headline size and of
    sa_hip_libsais(T, SA, 1e8), cold (workspace released: device buffers + pinned slabs are allocated inside the call) and
    warm, with the library's own breakdown and the PCIe floor of the bytes that have to cross (measured with pinned
    1 GiB copies on this box).  Never part of `value`."""
    out = {}
    n = text.size
    # PCIe floor: what a pinned copy of the same bytes takes
    pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    devbuf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    bw = {}
    for name, (dst, src) in (("h2d", (devbuf, pin)), ("d2h", (pin, devbuf))):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        bw[name] = (1 << 30) / (time.perf_counter() - t0)
    del pin, devbuf
    torch.cuda.empty_cache()
    out["pcie_pinned_gbps"] = {k: v / 1e9 for k, v in bw.items()}

    def run(label, fn, t, width, check):
        _capi.release_workspace()
        recs = []
        for kind in ("cold", "warm", "warm"):
            sa = np.empty(t.size, dtype=np.int64 if width == 8 else np.int32)   # untouched pages, as a caller's malloc would be
            t0 = time.perf_counter()
            rc = fn(t.ctypes.data, sa.ctypes.data, t.size, 0, None)
            dt = time.perf_counter() - t0
            assert rc == 0, rc
            b = _capi.last_call_breakdown()
            recs.append((kind, dt, b))
        floor = t.size / bw["h2d"] + 4 * t.size / bw["d2h"]
        warm = min(recs[1:], key=lambda r: r[1])
        out[label] = {"n_chars": int(t.size), "cold_s": recs[0][1], "warm_s": warm[1], "chars_per_s_warm": t.size / warm[1],
                      "pcie_floor_s": floor, "pcie_floor_note": "text up (n bytes) + suffix array down as u32 (4n bytes) at the pinned-copy rates above; "
                                                               "the int64 form is widened by host threads while the slabs arrive",
                      "breakdown_cold_ms": {k: recs[0][2][k] for k in ("workspace_ms", "upload_ms", "build_ms", "build_device_ms", "download_ms", "total_ms")},
                      "breakdown_warm_ms": {k: warm[2][k] for k in ("workspace_ms", "upload_ms", "build_ms", "build_device_ms", "download_ms", "total_ms")},
                      "equals_device_build": check(sa)}
        del sa

    lib = _capi.lib()
    run("sa_hip_libsais64_n1e9" if n == 1_000_000_000 else f"sa_hip_libsais64_n{n}", lib.sa_hip_libsais64, text, 8,
        lambda sa: bool(np.array_equal(sa, gpu_sa64)) if gpu_sa64 is not None else None)
    m = min(n, 100_000_000)
    t1 = np.ascontiguousarray(text[:m])
    if m == n and gpu_sa64 is not None:
        chk = lambda sa: bool(np.array_equal(sa, gpu_sa64.astype(np.int32)))
    else:
        with _capi.DeviceIndex(m, 0) as ix:
            ix.build(t1)
            ref32 = ix.sa_u32().view(np.int32).copy()
        chk = lambda sa: bool(np.array_equal(sa, ref32))
    run(f"sa_hip_libsais_n{m}", lib.sa_hip_libsais, t1, 4, chk)
    _capi.release_workspace()
    return out


def run_sharded(args, torch, dist, _capi, synth, rank, local_rank, world, dev):
    """BASELINE config 4 (see the module docstring)."""
    from suffixarray_amd.distributed import ShardedBatch, replicate_index, shard_bounds, slot_count
    N, Qg, m = args.n, args.queries_global, args.pattern_len
    text = synth.d1_uniform27(N)                     # the same text on every rank: rank 0 indexes it, all draw patterns from it

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    # construction: rank 0 only, once
    idx = _capi.DeviceIndex(N, local_rank)
    build_ms = None
    if rank == 0:
        idx.build(text)
        bms = []
        for _ in range(max(1, args.warmup) + 2):
            idx.build_device(idx.text_dev, N, 0)
            bms.append(idx.build_stats()["total_ms"])
        build_ms = min(bms[1:]) if len(bms) > 1 else bms[0]
    # replication: the builder's query structures (text, SA, key array, directory) by one RCCL broadcast per buffer
    # straight into buffers the replica has reserved; nothing is rebuilt there (sa_hip_index_replica_*).
    # (--exercise-dist: rank 0 too answers from a replica of its own index -- a device-to-device copy stands in for the
    # broadcast -- so that a single GPU runs every line a receiving rank runs.)
    searcher = idx
    dst = None
    if rank == 0 and args.exercise_dist:
        searcher = dst = _capi.DeviceIndex(N, local_rank)
    elif rank != 0:
        dst = idx
    barrier()
    b0 = time.perf_counter()
    bcast_bytes = replicate_index(idx if rank == 0 else None, dst, dev, src=0)
    searcher.sync()
    barrier()
    bcast_ms = (time.perf_counter() - b0) * 1e3

    # ONE global batch, the same on every rank; this rank's slice goes to its GPU
    lo, hi = shard_bounds(Qg, world, rank)
    q_buf, q_off = synth.query_batch(text, Qg, m, seed=0, lo=lo, hi=hi)
    # a chunk should carry enough patterns to pay for its collective (tens of microseconds of launch + rendezvous each): at most
    # --dist-chunks pieces, none under a million patterns -- 4 pieces at N <= 2, 2 at N = 4, 1 at N = 8 for the default batch
    # (from the rank-INDEPENDENT slot count: slices differ by one pattern, and ranks that disagreed on the chunk count would issue
    #  different numbers of collectives)
    chunks = max(1, min(args.dist_chunks, slot_count(Qg, world) // max(1, args.dist_min_chunk)))
    if args.dist_mode == "sharded_rows":
        return run_sharded_rows(args, torch, dist, synth, rank, world, dev, text, idx, searcher, build_ms, bcast_ms, bcast_bytes, (q_buf, q_off), barrier)
    batch = ShardedBatch(q_buf, q_off, Qg, world, rank, dev, chunks=chunks, mode=args.dist_mode,
                         search_stream=torch.cuda.ExternalStream(searcher.stream, device=dev))

    def search(pat_t, off_t, start, count, out_t):
        # asynchronous on the index's own stream; ShardedBatch orders it against the collectives with events
        if args.offsets_api:
            searcher.query_batch_device(pat_t.data_ptr(), off_t.data_ptr() + 8 * start, count, out_t.data_ptr())
        else:
            searcher.query_batch_device_fixed(pat_t.data_ptr() + m * start, m, count, out_t.data_ptr())

    for _ in range(args.warmup):
        batch.step(search)
    barrier()
    searcher.query_stats()   # drops the warm-up launches from the sums
    kern_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.step(search)
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = searcher.query_stats()["kernel_ms_sum"]
    tmax = torch.tensor([dt, kern_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt, kern_ms_max = tmax.tolist()

    # gate: the gathered table equals what rank 0's own (built) index answers for the whole batch
    gate = None
    one_gpu = None
    got = batch.results()
    if rank == 0:
        fb, fo = synth.query_batch(text, Qg, m, seed=0)
        exp = idx.query_batch((fb, fo))
        same = bool(np.array_equal(got["first"], exp["first"]) and np.array_equal(got["second"], exp["second"]))
        hits = (((got["second"].astype(np.int64) - got["first"].astype(np.int64) + 1) & 0xFFFFFFFF) > 0) & (got["first"] != 0xFFFFFFFF)
        gate = {"gathered_equals_single_gpu": same, "verify_violations": idx.verify(), "query_hit_rate": float(hits.mean())}
        gate["ok"] = bool(same and gate["verify_violations"] == 0)
        # the SAME global batch on ONE GPU (rank 0's built index, whole batch in one launch per step): the 1-GPU value of this
        # line's metric, so that strong scaling can be read off one line (the N = 1 bench line reports config 3, a build)
        pat_all = torch.from_numpy(np.concatenate([np.ascontiguousarray(fb, dtype=np.uint8), np.zeros(64, np.uint8)])).to(dev)
        off_all = torch.from_numpy(np.ascontiguousarray(fo, dtype=np.uint64).view(np.int64)).to(dev)
        out_all = torch.empty(2 * Qg, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        def whole():
            if args.offsets_api:
                idx.query_batch_device(pat_all.data_ptr(), off_all.data_ptr(), Qg, out_all.data_ptr())
            else:
                idx.query_batch_device_fixed(pat_all.data_ptr(), m, Qg, out_all.data_ptr())
        for _ in range(max(1, args.warmup)):
            whole()
        idx.sync()
        idx.query_stats()
        w0 = time.perf_counter()
        for _ in range(args.steps):
            whole()
        idx.sync()
        wdt = time.perf_counter() - w0
        one_gpu = {"queries_per_s": Qg * args.steps / wdt, "ms_per_step": wdt * 1e3 / args.steps,
                   "search_kernel_ms": idx.query_stats()["kernel_ms_sum"] / args.steps,
                   "note": "the whole batch on rank 0's GPU alone, ranges left in HBM (no gather): the 1-GPU reference for `value`"}
        del pat_all, off_all, out_all
        if args.dump:   # tests/test_gpu_dist.py: the gathered ranges and the suffix array, to be checked against the oracle
            np.savez(args.dump, first=got["first"], second=got["second"], sa=idx.sa_u32() if N <= 100_000_000 else np.zeros(0, np.uint32))
    steps = args.steps
    line = None
    if rank == 0:
        line = {
            "metric": "batched_queries_per_s",
            "value": Qg * steps / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8 text / u32 suffix indices / u32 range pairs",
            "data": "synthetic",
            "config": {"workload": f"config 4: D1 uniform27 text N={N:,}, ONE batch of {Qg:,} {m}-byte queries sharded over {world} GPU(s), "
                                   f"index built on rank 0 and replicated by RCCL broadcast, ranges gathered ({args.dist_mode}, {batch.chunks} chunks per step)",
                       "n_chars": N, "queries_global": Qg, "pattern_len": m,
                       "parallelism": f"replicated index, query batch sharded x{world} (no data-path collective; {args.dist_mode} of 8-byte ranges, "
                                      f"search of chunk k+1 overlapped with the gather of chunk k)"},
            "search_kernel_ms_max_rank": kern_ms_max / steps,
            "search_only_queries_per_s": Qg * steps / (kern_ms_max / 1e3) if kern_ms_max > 0 else None,
            "build_ms": build_ms,
            "build_chars_per_s": N / (build_ms / 1e3),
            "one_gpu_same_batch": one_gpu,
            "replicate_ms": bcast_ms,
            "replicate_bytes": bcast_bytes,
            "replicate_gbps": bcast_bytes / (bcast_ms / 1e3) / 1e9,
            "replicate_note": "layout + text + SA + key array + directory into reserved buffers, SA range check; nothing is rebuilt on the replica "
                              "(sa_hip_index_replica_*); at world size 1 a device-to-device copy stands in for the broadcast",
            "gate": gate,
        }
    if searcher is not idx:
        searcher.close()
    idx.close()
    return line


def run_sharded_rows(args, torch, dist, synth, rank, world, dev, text, idx, searcher, build_ms, bcast_ms, bcast_bytes, slice_pats, barrier):
    """--dist-mode sharded_rows: what a serving deployment does with a replicated index -- nothing of the per-query result
    crosses xGMI.  Every rank answers ITS slice completely (ranges + the distinct rows that contain each pattern, at most
    --rows-k per query: one search launch + one rows launch, sa_hip_index_query_rows_batch, host patterns in, host row ids
    out); one all-reduce of three counters per step is the only collective.  `value` = global queries/s."""
    N, Qg, m, k = args.n, args.queries_global, args.pattern_len, args.rows_k
    # rows of the D1 text = its lines; every rank derives the row table from the text it already holds
    starts = np.concatenate([[0], np.flatnonzero(text == 10).astype(np.uint64) + 1]).astype(np.uint64)
    if starts[-1] >= N:
        starts = starts[:-1]
    searcher.set_rows(starts)
    if searcher is not idx and rank == 0:
        idx.set_rows(starts)
    q_buf, q_off = slice_pats
    ql = q_off.size - 1

    # the result arrays of this rank's slice are the caller's and are kept from step to step, as a serving loop keeps them
    # (a fresh uint64[Q][k] per call costs ~60 ms of map / unmap work per step at Q = 1e7, k = 16: profiles/r04_n_rows_lanes.log)
    out = (np.empty((max(ql, 1), max(k, 1)), dtype=np.uint64), np.zeros(max(ql, 1), dtype=np.uint32),
           np.zeros(max(ql, 1), dtype=[("first", "<u4"), ("second", "<u4")]))

    def step():
        rows, ranges = searcher.query_rows_batch_raw((q_buf, q_off), k, out=out)
        c = torch.tensor([float(ql), float(rows[1].sum()), float((rows[1] > 0).sum())], dtype=torch.float64, device=dev)
        dist.all_reduce(c)
        return rows, ranges, c

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows, ranges, c = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    totals = c.tolist()
    line = None
    if rank == 0:
        fb, fo = synth.query_batch(text, Qg, m, seed=0)
        w0 = time.perf_counter()
        (wr, wc), wranges = idx.query_rows_batch_raw((fb, fo), k)
        wdt = time.perf_counter() - w0
        same = bool(int(totals[0]) == Qg and int(totals[1]) == int(wc.sum()) and int(totals[2]) == int((wc > 0).sum()))
        lo = 0
        live = np.arange(k)[None, :] < rows[1][:, None]          # row ids beyond a query's count are not written
        same = same and bool(np.array_equal(ranges["first"], wranges["first"][lo:lo + ql]) and np.array_equal(ranges["second"], wranges["second"][lo:lo + ql])
                             and np.array_equal(rows[1], wc[lo:lo + ql]) and np.array_equal(rows[0][live], wr[lo:lo + ql][live]))
        gate = {"sharded_equals_single_gpu": same, "verify_violations": idx.verify(), "rows_found": int(totals[1]),
                "queries_with_rows": int(totals[2])}
        gate["ok"] = bool(same and gate["verify_violations"] == 0)
        line = {
            "metric": "batched_queries_per_s", "value": Qg * args.steps / dt, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 text / u32 suffix indices / u32 range pairs / u64 row ids", "data": "synthetic",
            "config": {"workload": f"sharded_rows: D1 uniform27 text N={N:,} ({starts.size:,} rows), ONE batch of {Qg:,} {m}-byte queries sharded over {world} GPU(s), "
                                   f"index built on rank 0 and replicated; every rank returns ranges + up to {k} distinct row ids per query of ITS slice to its own host; "
                                   f"one all-reduce of three counters per step",
                       "n_chars": N, "queries_global": Qg, "pattern_len": m, "rows_k": k,
                       "parallelism": f"replicated index, query batch sharded x{world}, results stay sharded (no gather)"},
            "build_ms": build_ms, "build_chars_per_s": N / (build_ms / 1e3), "replicate_ms": bcast_ms, "replicate_bytes": bcast_bytes,
            "one_gpu_same_batch": {"queries_per_s": Qg / wdt, "ms_per_step": wdt * 1e3,
                                   "note": "the whole batch through the same call on rank 0's GPU alone (one run, after the timed loop)"},
            "gate": gate,
        }
    if searcher is not idx:
        searcher.close()
    idx.close()
    return line


if __name__ == "__main__":
    main()
