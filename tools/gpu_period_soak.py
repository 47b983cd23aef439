"""Randomized soak of round 3's build paths: the periodic-run shortcut (period_finish.hpp) and the 10-byte-record sort
(radix_narrow48.hpp) on and off (SA_HIP_PERIOD_FINISH x SA_HIP_NARROW48) over texts made of repeated blocks, runs, periodic
stretches and mutations over random alphabets, n up to 6e6 so that both sort plans are reached: identical suffix arrays, each
verified on the device; the int64 copy of a fused 64-bit build equal to the u32 array.
    python3 tools/gpu_period_soak.py [seed] [cases]"""
import os, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 99)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for c in range(cases):
    sigma = int(rng.choice([1, 2, 3, 4, 8, 26, 27, 60]))
    syms = rng.choice(np.arange(33, 250), sigma, replace=False).astype(np.uint8)
    n = int(rng.integers(300_000, 6_000_000))
    parts, total = [], 0
    blk = rng.choice(syms, int(rng.choice([1, 2, 3, 7, 50, 4_000, 90_000, 700_000])))
    while total < n:
        kind = rng.random()
        if kind < 0.55:
            p = np.tile(blk, int(rng.integers(2, 40)))                     # a periodic stretch
            if rng.random() < 0.3 and p.size > 10:
                p = p.copy(); p[rng.integers(0, p.size)] = syms[rng.integers(0, sigma)]   # one mutation: a run cut in two
        elif kind < 0.75:
            p = rng.choice(syms, int(rng.integers(10, 50_000)))
        elif kind < 0.9:
            p = np.full(int(rng.integers(10, 200_000)), syms[0], np.uint8)
        else:
            blk = rng.choice(syms, int(rng.choice([1, 2, 5, 13, 1_000, 30_000])))   # a new block from here on
            p = blk
        parts.append(p); total += p.size
    t = np.concatenate(parts)[:n]
    res = {}
    for mode in ("11", "10", "01", "00"):
        os.environ["SA_HIP_PERIOD_FINISH"] = mode[0]
        os.environ["SA_HIP_NARROW48"] = mode[1]
        with _capi.DeviceIndex(n, 0) as idx:
            idx.build(t)
            sa = idx.sa_u32().copy()
            v = idx.verify()
            st = idx.build_stats()
            ok64 = True
            if mode == "11":
                out = torch.full((n,), -7, dtype=torch.int64, device="cuda:0")
                torch.cuda.synchronize()
                idx.build_device64(idx.text_dev, n, out.data_ptr(), 0)
                idx.sync()
                ok64 = bool(np.array_equal(out.cpu().numpy(), sa.astype(np.int64)))
            res[mode] = (sa, v, st, ok64)
    ok = all(np.array_equal(res["11"][0], res[m][0]) and res[m][1] == 0 and res[m][3] for m in res)
    bad += not ok
    s1, s0 = res["11"][2], res["00"][2]
    print("case %2d sigma %2d n %7d: period-resolved %8d, narrow48 %d, rounds %2d vs %2d (dbl %2d vs %2d), %.1f vs %.1f ms -> %s" % (
        c, sigma, n, s1["period_resolved"], s1["narrow48"], s1["rounds"], s0["rounds"], s1["doubling_rounds"], s0["doubling_rounds"],
        s1["total_ms"], s0["total_ms"], "ok" if ok else "MISMATCH"), flush=True)
print("FAILED %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)
