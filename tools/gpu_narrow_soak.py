"""Randomized soak of the narrow-record sort: random alphabets / skews / sizes / key lengths (<= 40 bits), full and
truncated, text-sourced and key-sourced top-digit pass, each against the 12-byte-record plan and the device sufcheck."""
import os, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for c in range(cases):
    sigma = int(rng.choice([2, 3, 4, 5, 7, 8, 15, 16, 27, 31, 40, 63, 100, 200, 256]))
    n = int(rng.integers(4_200_000, 9_000_000))
    syms = rng.choice(256, sigma, replace=False).astype(np.uint8)
    p = rng.dirichlet(np.full(sigma, float(rng.choice([0.3, 1.0, 5.0]))))
    t = rng.choice(syms, n, p=p)
    if rng.random() < 0.3:   # long runs: big groups, deep refinement
        t[int(n * 0.4):int(n * 0.45)] = syms[0]
    b = int(np.ceil(np.log2(len(np.unique(t)) + 1)))
    kmax = 40 // b
    k0 = int(rng.integers(max(1, (9 + b - 1) // b), kmax + 1))
    L = int(rng.choice([0, 0, 3, 6, 32]))
    res = {}
    # query parity rides along: the narrow modes keep u32 narrow keys as the query key array, the plain one u64 keys
    pats = []
    for i in range(2000):
        m = int(rng.integers(1, 24))
        if i % 2 == 0:
            q = int(rng.integers(0, n - m))
            pats.append(bytes(t[q:q + m]))
        else:
            pats.append(bytes(rng.choice(syms, m)))
    for mode in ("text", "keys", "plain"):
        os.environ["SA_HIP_NARROW"] = "0" if mode == "plain" else "1"
        os.environ["SA_HIP_TEXT_PASS"] = "1" if mode == "text" else "0"
        os.environ["SA_HIP_INITIAL_CHARS"] = str(k0)
        with _capi.DeviceIndex(n, 0) as idx:
            idx.build(t, L)
            st = idx.build_stats()
            v = idx.verify()
            res[mode] = (idx.sa_u32().copy(), v, st, idx.query_batch(pats))
    same = np.array_equal(res["text"][0], res["plain"][0]) and np.array_equal(res["keys"][0], res["plain"][0])
    same = same and np.array_equal(res["text"][3], res["plain"][3]) and np.array_equal(res["keys"][3], res["plain"][3])
    used = res["text"][2]["pass_launches"][2] + res["text"][2]["pass_launches"][3]
    ok = same and all(res[m][1] == 0 for m in res)
    bad += not ok
    print("case %2d sigma %3d b %d k0 %2d L %2d n %8d narrow launches %d text_pass %d narrow_k %d rounds %2d split rb %2d / %2d max %6d flags %d -> %s" % (
        c, sigma, b, res["text"][2]["initial_chars"], L, n, used, res["text"][2]["text_top_pass"], res["text"][2]["narrow_k"],
        res["text"][2]["rounds"], res["text"][2]["split_plan"], res["keys"][2]["split_plan"], res["text"][2]["split_max"], res["text"][2]["lite_flags"],
        "ok" if ok else "MISMATCH"), flush=True)
print("FAILED %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)
