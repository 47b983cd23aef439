"""Where a single query_records call spends its time (config-5 style CSV index): python3 tools/gpu_query_latency.py [rows]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import SuffixArray, _capi
import ctypes as C

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
path = "/tmp/lat_%d.csv" % rows
_capi.synth_csv(path, rows, 1)
sa = SuffixArray(csv_file=path, search_column="company_name", max_suffix_length=32)
import csv
names = []
with open(path) as f:
    r = csv.reader(f); next(r)
    for i, row in enumerate(r):
        if i % (rows // 3000) == 0: names.append(row[1].upper())
        if len(names) >= 3000: break
def timed(fn, reps=1):
    ts = []
    for q in names:
        t0 = time.perf_counter(); fn(q); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    return "mean %.1f us median %.1f us p90 %.1f" % (ts.mean(), np.median(ts), np.percentile(ts, 90))
for k in (1000, 10, 1):
    print("query_records k=%-4d %s" % (k, timed(lambda q: sa.query_records(q, k=k))))
print("query_ranges (one pattern)  %s" % timed(lambda q: sa.query_ranges([q])))
# raw C call through ctypes: sa_hip_index_query_rows with k = 1000 (row ids only: no row copies, no Python shaping)
idx = sa._index
pats = [q.lower().encode() for q in names]
it = iter(pats)
print("C query_rows k=1000 (ctypes) %s" % timed(lambda q: idx.query_rows(q.lower().encode(), 1000)))
print("C query_rows k=1    (ctypes) %s" % timed(lambda q: idx.query_rows(q.lower().encode(), 1)))
print("C query_batch of ONE (ctypes) %s" % timed(lambda q: idx.query_batch([q.lower().encode()])))
print("C query_hits max 16 (ctypes)  %s" % timed(lambda q: idx.query_hits(q.lower().encode(), 16)))
print("len(names) =", len(names), " mean results =", np.mean([len(sa.query_records(q)) for q in names[:300]]))
os.remove(path)
