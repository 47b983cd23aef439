import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi
for n in [8192*8+1, 8192*9+1, 8192*100+1, 8192*1000+1, 8192*1000+63, 8192*1000+64, 8192*1000+65, 8192*1000+8191, 8192*8168+1, 8192*8168+2, 8192*2000+1]:
    rng = np.random.default_rng(n & 0xffff)
    keys = rng.integers(0, 1 << 47, n, dtype=np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    k, v = _capi.sort_pairs(keys, vals, 0, 47)
    order = np.argsort(keys, kind="stable")
    okk = np.array_equal(k, keys[order]); okv = np.array_equal(v, vals[order])
    uniq = np.unique(v).size
    print(n, n % 8192, "keys ok", okk, "vals ok", okv, "unique vals", uniq, "missing", n - uniq, flush=True)
    if not okv:
        bad = np.nonzero(v != vals[order])[0]
        print("   first bad slots", bad[:10], "count", bad.size, "tile of first", bad[0] // 8192, flush=True)
