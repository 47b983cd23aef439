"""The names batch all the way to row ids (bench.py: secondary.config5_csv_50M_rows.names_batch_1e6.rows_batch) on its own, A/B of
the forms of the rows kernels (csrc/rows_device.hpp) and of the host legs by SA_HIP_ROWS_LANES / SA_HIP_ROWS_WAVES / SA_HIP_ROWS_RING, every variant's rows
against the first (profiles/r04_o_names_rows.log also has the 1024-slot table that was measured and not kept):
    python3 tools/gpu_names_rows.py [rows] [queries] [k]"""
import os
import sys
import time

os.environ["SA_HIP_DIAG"] = "1"
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
path = "/tmp/companies_%d.csv" % rows
_capi.synth_csv(path, rows, 1)
col = np.array(_capi.csv_extract_column(path, "company_name", copy=False)[1])
os.remove(path)
rng = np.random.default_rng(0)
ends = np.flatnonzero(col == 10)
starts = np.concatenate([[0], ends[:-1] + 1]).astype(np.uint64)
pick = np.sort(rng.integers(1, ends.size, Q))
a, b = ends[pick - 1] + 1, ends[pick]
keep = b > a
a, b = a[keep], b[keep]
lens = (b - a).astype(np.uint64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
src = np.repeat(a - off[:-1].astype(np.int64), lens.astype(np.int64)) + np.arange(int(off[-1]), dtype=np.int64)
buf = col[src]
with _capi.DeviceIndex(col.size, 0) as idx:
    idx.build(col, 32)
    idx.set_rows(starts)
    idx.deep_keys(2)
    first = None
    for name, env in (("lanes + waves (default)", {}), ("SA_HIP_ROWS_WAVES=0", {"SA_HIP_ROWS_WAVES": "0"}), ("SA_HIP_ROWS_LANES=0", {"SA_HIP_ROWS_LANES": "0"}),
                      ("SA_HIP_ROWS_RING=0", {"SA_HIP_ROWS_RING": "0"}), ("wave groups 4", {"SA_HIP_ROWS_WAVE_GROUPS": "4"}), ("wave groups 16", {"SA_HIP_ROWS_WAVE_GROUPS": "16"}), ("wave groups 32", {"SA_HIP_ROWS_WAVE_GROUPS": "32"}), ("wave groups 64", {"SA_HIP_ROWS_WAVE_GROUPS": "64"}), ("lanes + waves again", {})):
        for k_, v_ in env.items():
            os.environ[k_] = v_
        out = (np.empty((a.size, K), dtype=np.uint64), np.zeros(a.size, dtype=np.uint32), np.zeros(a.size, dtype=_capi.PAIR_DTYPE))
        ms = []
        for _ in range(4):
            t0 = time.perf_counter()
            (rw, rc), rg = idx.query_rows_batch_raw((buf, off), K, out=out)
            ms.append((time.perf_counter() - t0) * 1e3)
        for k_ in env:
            os.environ.pop(k_)
        live = np.arange(K)[None, :] < rc[:, None]
        cur = (rw[live].copy(), rc.copy(), rg.copy())
        same = True if first is None else bool(np.array_equal(cur[0], first[0]) and np.array_equal(cur[1], first[1]) and np.array_equal(cur[2], first[2]))
        if first is None:
            first = cur
        print("%-42s call %.2f ms (best of 3 warm), %.1f M queries/s, mean rows %.2f, same as the first: %s" % (
            name, min(ms[1:]), a.size / min(ms[1:]) / 1e3, float(rc.mean()), same), flush=True)
        if not same:
            sys.exit(1)
print("ALL OK")
