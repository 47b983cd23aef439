"""One-off: index beyond INT32_MAX characters (32-bit unsigned suffix indices, n <= 2^32 - 2)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_000
t0 = time.time(); t = synth.d1_uniform27(n); print("gen %.1fs" % (time.time() - t0), flush=True)
with _capi.DeviceIndex(n, 0) as idx:
    t0 = time.time(); idx.build(t); print("build call %.2fs" % (time.time() - t0), idx.build_stats(), flush=True)
    t0 = time.time(); bad = idx.verify(); print("verify violations", bad, "%.2fs" % (time.time() - t0), flush=True)
    buf, off = synth.query_batch(t, 100000, 16)
    got = idx.query_batch((buf, off))
    pats = buf.reshape(-1, 16)
    ok = True
    rng = np.random.default_rng(0)
    for i in rng.integers(0, 100000, 300):
        f, s = int(got["first"][i]), int(got["second"][i])
        if f == 0xFFFFFFFF: continue
        p = pats[i].tobytes()
        if s >= f:
            pos = idx.sa_range(f, min(s - f + 1, 4))
            ok &= all(bytes(t[int(x):int(x) + 16]) == p for x in pos)
        else:
            a = idx.sa_range(f, 1)[0]
            ok &= bytes(t[int(a):int(a) + 16]) > p
            if f > 0:
                b = idx.sa_range(f - 1, 1)[0]
                ok &= bytes(t[int(b):int(b) + 16]) < p
    print("query spot checks ok:", ok, "hit rate", float((((got["second"].astype(np.int64) - got["first"].astype(np.int64) + 1) & 0xFFFFFFFF) > 0).mean()))
