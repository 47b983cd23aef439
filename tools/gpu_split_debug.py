"""Where does the three-pass plan go wrong?  python3 tools/gpu_split_debug.py [n]  (one box; diagnostic)
Builds D1 with SA_HIP_SPLIT=1, then looks at the suffix array whatever the return code: permutation? sorted by the top
8 / 8 + rb / 40 key bits?"""
import os, sys
os.environ["SA_HIP_DIAG"] = "1"
os.environ.setdefault("SA_HIP_SPLIT", "1")
os.environ["SA_HIP_TINY"] = os.environ.get("SA_HIP_TINY", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from suffixarray_amd import _capi as capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_500_001
t = synth.d1_uniform27(n)
idx = capi.DeviceIndex(t.size, 0)
try:
    idx.build(t)
    print("build ok", {k: idx.build_stats()[k] for k in ("split_plan", "split_max", "total_ms", "narrow_k", "lite_flags")})
except Exception as e:
    print("build failed:", e)
st = idx.build_stats()
print({k: st[k] for k in ("split_plan", "split_max", "initial_chars", "bits_per_symbol")})
sa = idx.sa_u32().astype(np.int64)
print("permutation:", np.array_equal(np.sort(sa), np.arange(n)))
# 40-bit keys of every slot
codes = np.zeros(256, np.int64); u = np.unique(t); codes[u] = np.arange(1, u.size + 1)
tp = np.concatenate([codes[t], np.zeros(16, np.int64)])
key = np.zeros(n, np.int64)
for c in range(8):
    key = (key << 5) | tp[sa + c]
for bits in (8, 8 + max(st["split_plan"], 1), 18, 29, 40):
    k = key >> (40 - bits)
    bad = np.flatnonzero(k[1:] < k[:-1])
    print("top %2d bits sorted: %s (%d inversions, first at %s)" % (bits, bad.size == 0, bad.size, bad[:3]))
