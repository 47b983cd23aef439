"""Soak: repeated builds with on-device verification (rare-race hunting)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth
rng = np.random.default_rng(1)
D1_REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cases = [("d1_1e9", synth.d1_uniform27(1_000_000_000), 0, D1_REPS),
         ("repeat_2^26", np.tile(rng.integers(97, 123, 1 << 20, dtype=np.uint8), 64), 0, 8),
         ("d2_2e8", synth.d2_words(200_000_000), 0, 8),
         ("d2_2e8_L32", None, 32, 8)]
prev = None
for name, t, L, reps in cases:
    if t is None: t = prev
    prev = t
    bad_total = 0
    with _capi.DeviceIndex(t.size, 0) as idx:
        t0 = time.time()
        for r in range(reps):
            idx.build(t, L)
            v = idx.verify()
            bad_total += v
            if v: print("  VIOLATIONS", name, "rep", r, v, idx.build_stats(), flush=True)
        print("%-12s n=%d L=%d reps=%d violations=%d (%.1fs) last build %.1f ms" % (name, t.size, L, reps, bad_total, time.time() - t0, idx.build_stats()["total_ms"]), flush=True)
