"""The names batch of bench.py (secondary.config5_csv_50M_rows.names_batch_1e6 / _8e6) on its own, for counter runs and A/B:
    python3 tools/gpu_names_batch.py [rows] [queries] [also_presorted: 1 / 0]
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/pmc_names -- python3 tools/gpu_names_batch.py [rows]
builds the company_name column index (L = 32) and answers 1e6 sampled names three times; prints kernel time and hit statistics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
SORTED = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
path = "/tmp/companies_%d.csv" % rows
_capi.synth_csv(path, rows, 1)
col = np.array(_capi.csv_extract_column(path, "company_name", copy=False)[1])
os.remove(path)
rng = np.random.default_rng(0)
ends = np.flatnonzero(col == 10)
pick = np.sort(rng.integers(1, ends.size, Q))
a, b = ends[pick - 1] + 1, ends[pick]
keep = b > a
a, b = a[keep], b[keep]
lens = (b - a).astype(np.uint64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
src = np.repeat(a - off[:-1].astype(np.int64), lens.astype(np.int64)) + np.arange(int(off[-1]), dtype=np.int64)
buf = col[src]
import time
with _capi.DeviceIndex(col.size, 0) as idx:
    idx.build(col, 32)
    idx.deep_keys(0)   # text comparisons inside a key group (round 3's search)
    ms0 = []
    for _ in range(3):
        res0 = idx.query_batch((buf, off))
        ms0.append(idx.query_stats()["kernel_ms"])
    t0 = time.perf_counter()
    has = idx.deep_keys(2)   # second-level keys (round 4)
    k2_ms = (time.perf_counter() - t0) * 1e3
    ms = []
    for _ in range(3):
        res = idx.query_batch((buf, off))
        ms.append(idx.query_stats()["kernel_ms"])
    # the same patterns in lexicographic order (what a caller -- or the library -- could do before the launch): lanes of a wave
    # then walk the same key groups
    if SORTED:
        pats = [bytes(col[x:y]) for x, y in zip(a, b)]
        order = sorted(range(len(pats)), key=pats.__getitem__)
        lens_s = lens[order]
        off_s = np.concatenate([[0], np.cumsum(lens_s)]).astype(np.uint64)
        buf_s = np.frombuffer(b"".join(pats[i] for i in order), np.uint8)
        ms_s = []
        for _ in range(3):
            res_s = idx.query_batch((buf_s, off_s))
            ms_s.append(idx.query_stats()["kernel_ms"])
        print("patterns sorted lexicographically: kernel %.3f ms (%.2f G queries/s); same ranges: %s" % (
            min(ms_s), a.size / min(ms_s) / 1e6, bool(np.array_equal(res_s, res[order]))))
    print("without deep keys: kernel %.3f ms (%.2f G queries/s); deep keys built=%s in %.1f ms (host clock); same ranges: %s" % (
        min(ms0), a.size / min(ms0) / 1e6, has, k2_ms, bool(np.array_equal(res, res0))))
cnt = ((res["second"].astype(np.int64) - res["first"].astype(np.int64) + 1) & 0xFFFFFFFF)
cnt[res["first"] == 0xFFFFFFFF] = 0
print("names batch: %d queries, mean length %.1f, kernel %.3f ms (%.2f G queries/s), hit rate %.3f, hits mean %.0f median %.0f p90 %.0f p99 %.0f" % (
    a.size, lens.mean(), min(ms), a.size / min(ms) / 1e6, (cnt > 0).mean(), cnt.mean(), np.median(cnt), np.percentile(cnt, 90), np.percentile(cnt, 99)))
# what phase 2 (binary search beyond the key) has to cover: the slots that share the pattern's first 11 characters
