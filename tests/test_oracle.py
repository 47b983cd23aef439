"""CPU: the oracle (own restatement) against the golden vectors produced by the reference, and
against the reference itself where oracle/_ref is built."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN
import cases


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_golden_small_sa(oracle):
    g = _load("golden_small.npz")
    for name in g["names"]:
        t, sa = g[f"text__{name}"], g[f"sa__{name}"]
        got = oracle.sais(t)
        assert np.array_equal(got, sa), name
        assert np.array_equal(oracle.sais64(t), sa.astype(np.int64)), name
        if t.size <= 5000:
            assert np.array_equal(oracle.sa_naive(t), sa.astype(np.uint32)), name
        assert oracle.sufcheck(t, sa.astype(np.uint32)) == 0


def test_golden_readme(oracle):
    g = _load("golden_readme.npz")
    t, sa = g["text"], g["sa"]
    assert np.array_equal(oracle.sais(t), sa)
    L = int(g["max_suffix_length"][0])
    pats = [bytes(p) for p in g["patterns"]]
    got = oracle.query_batch(t, sa.astype(np.uint32), L, pats)
    assert np.array_equal(got, g["ranges"])
    # "zzz" is greater than every suffix: the reference's not-found sentinel
    assert tuple(got[3]) == (0xFFFFFFFF, 0xFFFFFFFF)


def test_golden_1mb(oracle):
    g = _load("golden_1mb.npz")
    L = int(g["max_suffix_length"][0])
    for name in g["names"]:
        t = g[f"text__{name}"]
        sa = oracle.sais(t)
        assert hashlib.sha256(sa.astype("<i4").tobytes()).hexdigest() == str(g[f"sa_sha256__{name}"][0]), name
        pats = [bytes(p) for p in g[f"patterns__{name}"]]
        got = oracle.query_batch(t, sa.astype(np.uint32), L, pats)
        assert np.array_equal(got, g[f"ranges__{name}"]), name
        # the reference's own truncated SA yields the same ranges, and so does
        # the oracle's truncated order
        assert np.array_equal(g[f"ranges__{name}"], g[f"ranges_truncated_ref__{name}"]), name
        tsa = oracle.truncated_sa(t, L)
        assert np.array_equal(oracle.query_batch(t, tsa, L, pats), g[f"ranges__{name}"]), name


def check_truncated_order(t, sa, L):
    """sa is a permutation ordered by the first L bytes (a suffix that ends sorts first), ties in text order."""
    n = t.size
    assert np.array_equal(np.sort(sa), np.arange(n, dtype=np.uint32))
    if n < 2:
        return
    L = min(L, n)
    pad = np.concatenate([t.astype(np.int16) + 1, np.zeros(L, np.int16)])
    keys = np.stack([pad[sa.astype(np.int64) + j] for j in range(L)], axis=1)
    diff = keys[1:] != keys[:-1]
    anyd = diff.any(axis=1)
    first = diff.argmax(axis=1)
    rows = np.nonzero(anyd)[0]
    assert np.all(keys[1:][rows, first[rows]] > keys[:-1][rows, first[rows]])
    assert np.all(sa[1:][~anyd] > sa[:-1][~anyd])


def test_truncated_contract(oracle):
    for name, t in cases.small_texts().items():
        if t.size > 70000:
            continue
        for L in (1, 3, 8, 32):
            check_truncated_order(t, oracle.truncated_sa(t, L), L)


def test_oracle_vs_reference_build(oracle, ref):
    for name, t in cases.small_texts().items():
        if t.size < 1:
            continue
        assert np.array_equal(oracle.sais(t), ref.libsais(t, threads=1)), name
        assert np.array_equal(oracle.sais(t), ref.libsais(t, threads=4)), name
        if t.size > 1:
            assert np.array_equal(oracle.sais64(t), ref.libsais64(t, threads=2)), name


def test_oracle_vs_reference_query(oracle, ref):
    rng = np.random.default_rng(11)
    for sig in (26, 4, 2):
        n = 100_000
        t = (rng.integers(0, sig, n) + 97).astype(np.uint8)
        t[rng.random(n) < 0.04] = 10
        sa = oracle.sais(t).astype(np.uint32)
        tn = np.concatenate([t, np.zeros(64, np.uint8)])
        pats = []
        for i in range(1500):
            m = int(rng.integers(1, 41))
            if i % 2 == 0:
                p = int(rng.integers(0, n - m))
                pats.append(bytes(t[p:p + m]).replace(b"\n", b"a"))
            else:
                pats.append(bytes((rng.integers(0, sig, m) + 97).astype(np.uint8)))
        got = oracle.query_batch(t, sa, 32, pats)
        for q, g in zip(pats, got):
            assert tuple(g) == ref.query(tn, sa, 32, q), q


def test_query_edge_conventions(oracle):
    t = np.frombuffer(b"banana\nbandana\n", np.uint8)
    sa = oracle.sais(t).astype(np.uint32)
    n = t.size
    assert oracle.query(t, sa, 32, b"") == (0, n - 1)                   # empty pattern matches all
    assert oracle.query(t, sa, 32, b"zz") == (0xFFFFFFFF, 0xFFFFFFFF)    # greater than every suffix
    f, s = oracle.query(t, sa, 32, b"ban")
    assert s - f + 1 == 2
    f, s = oracle.query(t, sa, 32, b"bab")                               # miss: first = lb, last = lb - 1
    assert s == f - 1
    f, s = oracle.query(t, sa, 32, b"\x01")                              # smaller than every suffix
    assert (f, s) == (0, 0xFFFFFFFF)
    assert oracle.query(t, sa, 2, b"baXYZ") == oracle.query(t, sa, 32, b"ba")  # truncated to L
