"""GPU: seeded randomized differential test -- many small texts of varied structure (alphabet size,
runs, periodicity, repeats, NUL / 0xFF bytes) through the C ABI against the oracle: full suffix array,
truncated order for a random L, and query ranges."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_text(rng):
    n = int(rng.choice([1, 2, 3, 7, 64, 65, 300, 1000, 4097, 9000]))
    kind = rng.integers(0, 6)
    sigma = int(rng.choice([1, 2, 3, 4, 16, 27, 100, 256]))
    if kind == 0:
        t = rng.integers(0, sigma, n)
    elif kind == 1:                               # runs
        t = np.repeat(rng.integers(0, sigma, n // 3 + 1), rng.integers(1, 9, n // 3 + 1))[:n]
    elif kind == 2:                               # periodic
        p = int(rng.integers(1, 40))
        t = np.tile(rng.integers(0, sigma, p), n // p + 1)[:n]
    elif kind == 3:                               # repeated block with mutations
        blk = rng.integers(0, sigma, max(1, n // 5))
        t = np.tile(blk, 6)[:n].copy()
        t[rng.integers(0, n, max(1, n // 50))] = rng.integers(0, sigma, max(1, n // 50))
    elif kind == 4:                               # text-like
        words = [rng.integers(97, 123, rng.integers(1, 8)) for _ in range(30)]
        parts, total = [], 0
        while total < n:
            parts.append(words[int(rng.integers(0, 30))])
            total += len(parts[-1]) + 1
        t = np.concatenate([np.concatenate([w, [32]]) for w in parts])[:n]
    else:                                         # extreme byte values
        t = rng.choice(np.array([0, 1, 127, 128, 254, 255]), n)
    t = np.asarray(t, dtype=np.uint8)
    if t.size < n:
        t = np.concatenate([t, np.zeros(n - t.size, np.uint8)])
    return t


def test_randomized_differential(gpu, oracle):
    rng = np.random.default_rng(20260101)
    with gpu.DeviceIndex(9000, 0) as idx:
        for case in range(250):
            t = _random_text(rng)
            exp = oracle.sais(t).astype(np.uint32)
            idx.build(t)
            got = idx.sa_u32()
            assert np.array_equal(got, exp), (case, t[:40].tolist(), idx.build_stats())
            # queries on the full index
            pats = []
            for _ in range(12):
                m = int(rng.integers(0, 45))
                if rng.random() < 0.6 and t.size > m:
                    p = int(rng.integers(0, t.size - m + 1))
                    pats.append(bytes(t[p:p + m]))
                else:
                    pats.append(bytes(rng.integers(0, 256, m, dtype=np.uint8)))
            assert np.array_equal(idx.query_batch(pats), oracle.query_batch(t, exp, 0xFFFFFFFF, pats)), case
            # truncated order + queries truncated at L
            L = int(rng.choice([1, 2, 3, 5, 8, 13, 32, 64]))
            idx.build(t, L)
            tsa = idx.sa_u32()
            texp = oracle.truncated_sa(t, L)
            assert np.array_equal(tsa, texp), (case, L, t[:40].tolist(), idx.build_stats())
            assert np.array_equal(idx.query_batch(pats), oracle.query_batch(t, texp, L, pats)), (case, L)
