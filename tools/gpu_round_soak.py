"""Randomized soak of the tile-local round sort (round_sort.hpp): texts made of repeated, mutated segments over random
alphabets -- groups of every size, long common prefixes, chunk and doubling rounds, truncation depths -- built with
SA_HIP_LOCAL_ROUNDS = 1 and 0: identical suffix arrays, each verified on the device."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4242)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for c in range(cases):
    sigma = int(rng.choice([2, 3, 4, 8, 20, 27, 64, 200]))
    syms = rng.choice(256, sigma, replace=False).astype(np.uint8)
    n = int(rng.integers(200_000, 3_000_000))
    seg_len = int(rng.choice([50, 700, 5_000, 40_000, 300_000]))
    seg = rng.choice(syms, seg_len)
    parts, total = [], 0
    while total < n:
        kind = rng.random()
        if kind < 0.5:
            p = seg.copy()
            k = int(rng.integers(0, 4))
            if k:
                p[rng.integers(0, seg_len, k)] = rng.choice(syms, k)   # a few mutations: long but finite common prefixes
        elif kind < 0.8:
            p = rng.choice(syms, int(rng.integers(10, 20_000)))
        else:
            p = np.full(int(rng.integers(10, 5_000)), syms[0], np.uint8)  # runs
        parts.append(p); total += p.size
    t = np.concatenate(parts)[:n]
    L = int(rng.choice([0, 0, 0, 9, 40, 200]))
    res = {}
    for mode in ("1", "0"):
        os.environ["SA_HIP_LOCAL_ROUNDS"] = mode
        with _capi.DeviceIndex(n, 0) as idx:
            idx.build(t, L)
            res[mode] = (idx.sa_u32().copy(), idx.verify(), idx.build_stats())
    ok = np.array_equal(res["1"][0], res["0"][0]) and res["1"][1] == 0 and res["0"][1] == 0
    bad += not ok
    s1, s0 = res["1"][2], res["0"][2]
    print("case %2d sigma %3d n %7d seg %6d L %3d rounds %2d (chunk %d dbl %d) global passes %3d vs %3d -> %s" % (
        c, sigma, n, seg_len, L, s1["rounds"], s1["chunk_rounds"], s1["doubling_rounds"], s1["radix_passes"], s0["radix_passes"],
        "ok" if ok else "MISMATCH"), flush=True)
print("FAILED %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)
