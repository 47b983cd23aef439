"""End-to-end time of SuffixArray(documents=...) for many short documents: python3 tools/gpu_docs_e2e.py [docs]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from suffixarray_amd import SuffixArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
rng = np.random.default_rng(3)
words = ["acme", "globex", "initech", "umbrella", "hooli", "vehement", "massive", "dynamic", "stark", "wayne", "wonka", "inc", "llc", "ltd"]
idx = rng.integers(0, len(words), (n, 3))
docs = [words[a] + " " + words[b] + " " + words[c] for a, b, c in idx]
torch.cuda.init(); torch.zeros(1, device="cuda")
for rep in range(2):
    t0 = time.perf_counter()
    sa = SuffixArray(documents=docs, max_suffix_length=32)
    t1 = time.perf_counter()
    qt = []
    for q in ("stark wonka", "acme inc", "hooli", "zzz"):
        ta = time.perf_counter(); r = sa.query_records(q, k=5); qt.append((time.perf_counter() - ta) * 1e6)
    print("rep %d: %d documents, construction %.3f s (device build %.1f ms), first queries %s us" % (
        rep, n, t1 - t0, sa._index.build_stats()["total_ms"], " / ".join("%.0f" % x for x in qt)), flush=True)
    sa.close()
