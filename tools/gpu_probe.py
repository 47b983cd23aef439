"""Staged on-device sanity run with verbose diagnostics (first thing to run on a GPU box)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth
from oracle.oracle import Oracle

o = Oracle()
print("devices", _capi.lib().sa_hip_device_count(), flush=True)

def first_diff(a, b):
    d = np.nonzero(a != b)[0]
    return None if d.size == 0 else (int(d[0]), int(d.size))

# 1. sort alone
for n in (1000, 5000, 100000, 3_000_000):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 62, n, dtype=np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    t0 = time.time()
    k, v = _capi.sort_pairs(keys, vals)
    order = np.argsort(keys, kind="stable")
    print("sort", n, "keys", first_diff(k, keys[order]), "vals", first_diff(v, vals[order]), "%.3fs" % (time.time() - t0), flush=True)

# 2. builds
for name, t in [("banana", np.frombuffer(b"banana", np.uint8)), ("d1_1e3", synth.d1_uniform27(1000)),
                ("d1_1e5", synth.d1_uniform27(100_000)), ("all_a", synth.all_same(20000)),
                ("fib", synth.fibonacci(30000)), ("d2_1e6", synth.d2_words(1_000_000)), ("d1_1e7", synth.d1_uniform27(10_000_000))]:
    with _capi.DeviceIndex(t.size, 0) as idx:
        t0 = time.time()
        idx.build(t)
        sa = idx.sa_u32()
        dt = time.time() - t0
        exp = o.sais(t).astype(np.uint32)
        print("build", name, t.size, "diff", first_diff(sa, exp), "%.3fs" % dt, idx.build_stats(), flush=True)
        pats = [bytes(t[p:p + 8]) for p in range(0, max(t.size - 8, 1), max(t.size // 50, 1))] + [b"zz", b"a", b""]
        got = idx.query_batch(pats)
        ex = o.query_batch(t, exp, 0xFFFFFFFF, pats)
        print("query", name, "diff", first_diff(got["first"], ex["first"]), first_diff(got["second"], ex["second"]), flush=True)
print("probe done")
