"""Randomized soak of the THREE-PASS plan of the narrow sort (csrc/radix_split.hpp): near-uniform texts -- the only ones
that take the plan -- over random alphabets, sizes (4.2e6 .. 3e8: levels rb = 0 .. 9, both forms of the local pass), key
lengths, truncation lengths and bounds on a sub-bucket; every case is built with the plan (flags work folded into the local
pass) and with the LSD passes (SA_HIP_SPLIT=0) and must give the same suffix array and the same query ranges, 0 violations
in the device sufcheck.  A mildly uneven distribution rides along (the plan must either hold or be declined -- never wrong).

    python3 tools/gpu_split_soak.py [seed] [cases] [max_n]
"""
import os, sys, time
os.environ.setdefault("SA_HIP_DIAG", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 31337)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
max_n = float(sys.argv[3]) if len(sys.argv) > 3 else 3e8
bad = 0
taken = 0
levels = set()
for c in range(cases):
    sigma = int(rng.choice([4, 5, 8, 16, 20, 27, 32, 33, 64, 100, 128, 256]))
    n = int(np.exp(rng.uniform(np.log(4.2e6), np.log(max_n))))
    syms = np.sort(rng.choice(256, sigma, replace=False)).astype(np.uint8)
    shape = rng.random()
    if shape < 0.6:        # exactly uniform
        t = syms[rng.integers(0, sigma, n, dtype=np.uint8 if sigma < 256 else np.uint16)]
    else:                  # mildly uneven: symbol probabilities within a few percent (60 %) or tens of percent (40 %) of each other
        a = float(rng.choice([2000.0, 2000.0, 2000.0, 60.0, 60.0]))
        p = rng.dirichlet(np.full(sigma, a))
        cdf = np.cumsum(p)
        cdf[-1] = 1.0
        t = syms[np.searchsorted(cdf, rng.random(n, dtype=np.float32), side="right").clip(0, sigma - 1)]
    t = np.ascontiguousarray(t, dtype=np.uint8)
    L = int(rng.choice([0, 0, 0, 8, 32]))
    cap = int(rng.choice([0, 0, 0, 300, 1000, 2048, 5000, 7000]))
    b = int(np.ceil(np.log2(sigma + 1)))
    k0 = int(rng.choice([0, 0, 0, max(2, 40 // b - 1), max(2, 40 // b - 2)]))
    env = {"SA_HIP_INITIAL_CHARS": str(k0) if k0 else None, "SA_HIP_SPLIT_CAP": str(cap) if cap else None}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    pats = []
    for i in range(3000):
        m = int(rng.integers(1, 24))
        if i % 2 == 0:
            q = int(rng.integers(0, n - m))
            pats.append(bytes(t[q:q + m]))
        else:
            pats.append(bytes(syms[rng.integers(0, sigma, m)]))
    pats += [b"", bytes(t[-5:]), bytes(t[-1:]), bytes(t[:7]), bytes([255]) * 3]
    res = {}
    t0 = time.time()
    for mode in ("F", "0"):
        os.environ["SA_HIP_SPLIT"] = "0" if mode == "0" else "1"
        os.environ["SA_HIP_SPLIT_FLAGS"] = "1"
        with _capi.DeviceIndex(n, 0) as idx:
            idx.build(t, L)
            st = idx.build_stats()
            v = idx.verify()
            res[mode] = (idx.sa_u32().copy(), v, st, idx.query_batch(pats).copy())
    same = np.array_equal(res["F"][0], res["0"][0]) and np.array_equal(res["F"][3], res["0"][3])
    ok = same and res["F"][1] == 0 and res["0"][1] == 0 and res["0"][2]["split_plan"] == 0
    bad += not ok
    st = res["F"][2]
    taken += st["split_plan"] > 0
    levels.add(st["split_plan"])
    print("case %2d sigma %3d n %9d L %2d cap %5d k0 %2d (forced %2d) narrow_k %d rounds %d split rb %2d max %7d ms %7.2f / %7.2f  %4.1f s -> %s" % (
        c, sigma, n, L, cap, st["initial_chars"], k0, st["narrow_k"], st["rounds"], st["split_plan"], st["split_max"],
        st["total_ms"], res["0"][2]["total_ms"], time.time() - t0, "ok" if ok else "MISMATCH"), flush=True)
    del res, t
print("plan taken in %d of %d cases, levels %s" % (taken, cases, sorted(levels)))
print("FAILED %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)
