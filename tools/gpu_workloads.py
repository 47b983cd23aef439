"""Build timing + verification over several text distributions (diagnostic)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth

def names(n, seed=5):
    rng = np.random.default_rng(seed)
    vocab = [bytes(rng.integers(97, 123, rng.integers(3, 10), dtype=np.uint8)) for _ in range(20000)]
    suffixes = [b" inc", b" llc", b" ltd", b", inc.", b"", b"", b""]
    out = bytearray()
    w = 1.0 / np.arange(1, len(vocab) + 1); cdf = np.cumsum(w / w.sum())
    while len(out) < n:
        k = rng.integers(1, 4)
        ids = np.searchsorted(cdf, rng.random(k))
        out += b" ".join(vocab[i] for i in ids) + suffixes[rng.integers(0, len(suffixes))] + b"\n"
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None
rng = np.random.default_rng(1)
cases = [
    ("d1_uniform27", lambda: synth.d1_uniform27(N), 0),
    ("d2_words", lambda: synth.d2_words(N), 0),
    ("dna4", lambda: (rng.integers(0, 4, N, dtype=np.uint8) + 97), 0),
    ("bytes256", lambda: rng.integers(0, 256, N, dtype=np.uint8), 0),
    ("names_full", lambda: names(N), 0),
    ("names_L32", lambda: names(N), 32),
    ("d1_L32", lambda: synth.d1_uniform27(N), 32),
    ("repeat_1MB_block", lambda: np.tile(rng.integers(97, 123, 1 << 20, dtype=np.uint8), N // (1 << 20) + 1)[:N].copy(), 0),
    ("all_a_1e7", lambda: synth.all_same(10_000_000), 0),
    ("fib_1e7", lambda: synth.fibonacci(10_000_000), 0),
]
for name, gen, L in cases:
    if ONLY and name not in ONLY:
        continue
    t = gen()
    with _capi.DeviceIndex(t.size, 0) as idx:
        idx.build(t, L)           # warm-up (allocations)
        idx.build(t, L)
        st = idx.build_stats()
        t0 = time.time(); bad = idx.verify(); tv = time.time() - t0
        print("%-18s n=%-11d L=%-3d total %8.2f ms (%6.2f Gchars/s) radix %8.2f ms passes %3d k0=%2d b=%d rounds %d (chunk %d, dbl %d) depth %d active_total %d verify=%d (%.0f ms)" % (
            name, t.size, L, st["total_ms"], t.size / st["total_ms"] / 1e6, st["radix_ms"], st["radix_passes"], st["initial_chars"],
            st["bits_per_symbol"], st["rounds"], st["chunk_rounds"], st["doubling_rounds"], st["final_depth"], st["active_total"], bad, tv * 1e3),
            "period_resolved", st.get("period_resolved"), flush=True)
