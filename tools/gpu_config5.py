"""BASELINE config 5: CSV-indexer mode -- 50M-row synthetic company_name column, max_suffix_length = 32,
build + query_records (reference protocol tests/test.py:99-141: 10 000 sampled names, mean/median us)."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, SuffixArray
from csv_ingest import extract_column

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
path = "/tmp/companies_%d.csv" % rows
t0 = time.time(); _capi.synth_csv(path, rows, 1); t_gen = time.time() - t0
size = os.path.getsize(path)
t0 = time.time(); col = extract_column(path, "company_name"); t_ext = time.time() - t0
n = len(col.text)
print("rows %d file %.2f GB gen %.1fs | extract %.1fs (%.0f MB/s) column text %d chars" % (rows, size / 1e9, t_gen, t_ext, size / 1e6 / t_ext, n), flush=True)
t0 = time.time()
s = SuffixArray(csv_file=path, search_column="company_name", max_suffix_length=32)
t_total = time.time() - t0
idx = s._index
st = idx.build_stats()
print("SuffixArray(csv_file) total %.1fs; device build %.1f ms (%.2f Gchars/s), rounds %d, passes %d, k0=%d, active_total %d" % (
    t_total, st["total_ms"], n / st["total_ms"] / 1e6, st["rounds"], st["radix_passes"], st["initial_chars"], st["active_total"]), flush=True)
idx.build(col.text, 32)   # second build: allocations warm
st = idx.build_stats()
print("warm device build %.1f ms (%.2f Gchars/s) verify=%d" % (st["total_ms"], n / st["total_ms"] / 1e6, idx.verify()), flush=True)
names = col.text.split(b"\n")[:-1]
rng = np.random.default_rng(0)
sample = [names[i].decode() for i in rng.integers(0, len(names), 10_000)]
lat, nres = [], []
for q in sample:
    t0 = time.perf_counter(); r = s.query_records(q.upper()); lat.append((time.perf_counter() - t0) * 1e6); nres.append(len(r))
lat = np.array(lat)
print("query_records: 10000 sampled names: mean %.1f us median %.1f us mean #results %.1f" % (lat.mean(), np.median(lat), np.mean(nres)), flush=True)
# batched ranges
big = [names[i] for i in rng.integers(0, len(names), 1_000_000)]
buf = b"".join(big); off = np.zeros(len(big) + 1, np.uint64); off[1:] = np.cumsum([len(b) for b in big])
t0 = time.perf_counter(); res = idx.query_batch((np.frombuffer(buf, np.uint8), off)); t_b = time.perf_counter() - t0
print("query_batch 1e6 names (host buffers incl. PCIe): %.1f ms; kernel %.3f ms (%.2f G queries/s)" % (t_b * 1e3, idx.query_stats()["kernel_ms"], 1e6 / idx.query_stats()["kernel_ms"] / 1e6), flush=True)
hits = ((res["second"].astype(np.int64) - res["first"].astype(np.int64) + 1) & 0xFFFFFFFF)
print("all found:", bool((hits > 0).all()), "mean hits", hits.mean())
os.remove(path)
