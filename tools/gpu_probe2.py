import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth
rng = np.random.default_rng(1)
for blk_log, n in [(10, 1 << 16), (12, 1 << 18), (14, 1 << 20), (16, 1 << 22), (20, 1 << 24), (20, 1 << 26), (4, 1 << 22)]:
    blk = rng.integers(97, 123, 1 << blk_log, dtype=np.uint8)
    t = np.tile(blk, n // blk.size + 1)[:n].copy()
    with _capi.DeviceIndex(t.size, 0) as idx:
        try:
            idx.build(t)
            print("ok  ", blk_log, n, "verify", idx.verify(), idx.build_stats(), flush=True)
        except Exception as e:
            print("FAIL", blk_log, n, e, idx.build_stats(), flush=True)
for name, t in [("all_a_1e6", synth.all_same(1_000_000)), ("all_a_1e7", synth.all_same(10_000_000)), ("fib_1e7", synth.fibonacci(10_000_000))]:
    with _capi.DeviceIndex(t.size, 0) as idx:
        try:
            idx.build(t)
            print("ok  ", name, "verify", idx.verify(), idx.build_stats(), flush=True)
        except Exception as e:
            print("FAIL", name, e, idx.build_stats(), flush=True)
