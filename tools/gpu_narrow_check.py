"""Narrow-record initial sort (radix_narrow.hpp) against the plain 12-byte-record sort: same SA, verified."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth

rng = np.random.default_rng(3)
def skewed(n):   # one very large bucket, many empty ones, 40-bit keys
    t = rng.choice(np.array([97, 98, 99, 100, 122], dtype=np.uint8), n, p=[0.9, 0.04, 0.03, 0.02, 0.01])
    return t
def two(n):      # two symbols: b = 2, forced 20 characters -> 40 bits, 4 buckets in use... most of the 256 empty
    return rng.choice(np.array([97, 122], dtype=np.uint8), n)
# (name, generator, n, forced initial characters or 0)
cases = [("d1", synth.d1_uniform27, 4_500_001, 0), ("d1", synth.d1_uniform27, 20_000_000, 0), ("skew5_k13", skewed, 30_000_000, 13),
         ("skew5_k9", skewed, 12_345_678, 9), ("two_k20", two, 16_000_000, 20), ("words_k8", synth.d2_words, 40_000_000, 8),
         ("d1_k7", synth.d1_uniform27, 25_000_000, 7), ("d1", synth.d1_uniform27, 100_000_000, 0), ("d1_L32", synth.d1_uniform27, 50_000_000, 0)]
ok = True
for name, gen, n, k0 in cases:
    t = gen(n)
    L = 32 if name.endswith("L32") else 0
    res = {}
    for mode in ("1", "t0", "0"):
        os.environ["SA_HIP_NARROW"] = "0" if mode == "0" else "1"
        os.environ["SA_HIP_TEXT_PASS"] = "0" if mode == "t0" else "1"
        if k0: os.environ["SA_HIP_INITIAL_CHARS"] = str(k0)
        else: os.environ.pop("SA_HIP_INITIAL_CHARS", None)
        with _capi.DeviceIndex(t.size, 0) as idx:
            idx.build(t, L)
            idx.build(t, L)
            st = idx.build_stats()
            bad = idx.verify()
            res[mode] = (idx.sa_u32().copy(), st, bad)
            print("%-8s n=%-10d L=%-2d narrow=%s total %7.2f ms radix %7.2f ms passes %2d bytes/rec %.1f k0=%d b=%d verify=%d" % (
                name, n, L, mode, st["total_ms"], st["radix_ms"], st["radix_passes"], st["radix_bytes"] / n, st["initial_chars"],
                st["bits_per_symbol"], bad), flush=True)
    same = np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["t0"][0], res["0"][0])
    print("   same SA:", same, flush=True)
    ok &= same and res["1"][2] == 0 and res["0"][2] == 0 and res["t0"][2] == 0
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
