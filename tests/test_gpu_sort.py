"""GPU: the one-sweep radix sort on its own (sa_hip_sort_pairs) against numpy's stable argsort."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 4095, 4096, 4097, 8192, 100_000, 1_000_003])
def test_sort_pairs_matches_stable_argsort(gpu, n):
    rng = np.random.default_rng(n)
    for bits in ((0, 64), (0, 8), (24, 64), (4, 37), (60, 64)):
        keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
        if bits == (0, 8):
            keys &= np.uint64(0xFFFF)   # heavy duplicates: stability matters
        vals = rng.integers(0, 1 << 32, n, dtype=np.uint32)
        lo, hi = bits
        mask = np.uint64(((1 << (hi - lo)) - 1) << lo) if hi - lo < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
        order = np.argsort(keys & mask, kind="stable")
        k, v = gpu.sort_pairs(keys, vals, lo, hi)
        assert np.array_equal(k, keys[order]), (n, bits)
        assert np.array_equal(v, vals[order]), (n, bits)


def test_sort_skewed_digits(gpu):
    n = 300_000
    keys = np.zeros(n, dtype=np.uint64)
    keys[::7] = 1 << 40
    keys[::13] = 0xFFFFFFFFFFFFFFFF
    vals = np.arange(n, dtype=np.uint32)
    order = np.argsort(keys, kind="stable")
    k, v = gpu.sort_pairs(keys, vals)
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])
