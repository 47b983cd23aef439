"""Randomized soak of the refinement machinery (group_finish.hpp, round_sort.hpp): texts made of repeated, mutated segments
over random alphabets -- groups of every size, long common prefixes, chunk and doubling rounds, truncation depths -- built
with the in-LDS group finisher on and off and the tile-local round sort on and off (SA_HIP_GROUP_FINISH x
SA_HIP_LOCAL_ROUNDS), with random initial key lengths: identical suffix arrays, each verified on the device.
    python3 tools/gpu_round_soak.py [seed] [cases]"""
import os, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
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
    k0 = int(rng.choice([0, 0, 2, 4, 7]))
    if k0:
        os.environ["SA_HIP_INITIAL_CHARS"] = str(k0)
    else:
        os.environ.pop("SA_HIP_INITIAL_CHARS", None)
    for mode in ("11", "10", "01", "00"):
        os.environ["SA_HIP_GROUP_FINISH"] = mode[0]
        os.environ["SA_HIP_LOCAL_ROUNDS"] = mode[1]
        with _capi.DeviceIndex(n, 0) as idx:
            idx.build(t, L)
            res[mode] = (idx.sa_u32().copy(), idx.verify(), idx.build_stats())
    ok = all(np.array_equal(res["11"][0], res[m][0]) and res[m][1] == 0 for m in res)
    bad += not ok
    s1, s0 = res["11"][2], res["00"][2]
    print("case %2d sigma %3d n %7d seg %6d L %3d k0 %d: finisher resolved %8d of %8d looked at in %d runs; rounds %2d vs %2d (dbl %d vs %d) -> %s" % (
        c, sigma, n, seg_len, L, k0, s1["finisher_resolved"], s1["finisher_records"], s1["finisher_runs"], s1["rounds"], s0["rounds"],
        s1["doubling_rounds"], s0["doubling_rounds"], "ok" if ok else "MISMATCH"), flush=True)
print("FAILED %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)
