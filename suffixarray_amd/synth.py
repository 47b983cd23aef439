"""Synthetic inputs of SURVEY.md 8(d): D1 uniform27, D2 words, D3 adversarial, query batches.
Used by bench.py and the tests (numpy only; D1 uses the C generator in libsa_hip.so when the
library is built, else an identical pure-Python loop)."""
import numpy as np

D1_SEED = 88172645463325252
_M64 = (1 << 64) - 1


def d1_uniform27(n, seed=D1_SEED):
    """xorshift64 (13,7,17); symbol = (s >> 33) % 27; 26 -> '\\n'."""
    try:
        from . import _capi
        return _capi.synth_uniform27(n, seed)
    except ImportError:
        out = np.empty(n, dtype=np.uint8)
        s = seed
        for i in range(n):
            s ^= (s << 13) & _M64
            s ^= s >> 7
            s ^= (s << 17) & _M64
            v = (s >> 33) % 27
            out[i] = 10 if v == 26 else 97 + v
        return out


def d2_words(n, seed=12345, vocab=50000):
    """Zipf(1.0) draws from a synthetic lower-case vocabulary (word length 2..12), space
    separated, '\\n' every 8..16 words."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(2, 13, vocab)
    letters = rng.integers(97, 123, int(lens.sum()), dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    w = 1.0 / np.arange(1, vocab + 1)
    cdf = np.cumsum(w / w.sum())
    avg = float((lens * (w / w.sum())).sum()) + 1.0
    nwords = int(n / avg * 1.15) + 64
    ids = np.searchsorted(cdf, rng.random(nwords)).clip(0, vocab - 1)
    wl = lens[ids]
    tot = int((wl + 1).sum())
    out = np.full(tot, 32, dtype=np.uint8)
    pos = np.concatenate([[0], np.cumsum(wl + 1)[:-1]])
    # scatter the letters of every word
    maxl = int(lens.max())
    for j in range(maxl):
        m = wl > j
        out[pos[m] + j] = letters[starts[ids[m]] + j]
    # newline instead of space every 8..16 words
    k = 0
    gaps = rng.integers(8, 17, nwords // 8 + 2)
    idx = np.cumsum(gaps)
    idx = idx[idx < nwords]
    out[pos[idx] + wl[idx]] = 10
    del k
    return out[:n].copy()


def _d2_tables(seed, vocab):
    rng = np.random.default_rng(seed)
    lens = rng.integers(2, 13, vocab)
    letters = rng.integers(97, 123, int(lens.sum()), dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    w = 1.0 / np.arange(1, vocab + 1)
    p = w / w.sum()
    return lens, letters, starts, np.cumsum(p), float((lens * p).sum()) + 1.0


def _d2_part(n, tabs, seed, k):
    lens, letters, starts, cdf, avg = tabs
    rng = np.random.default_rng([seed, k])
    nwords = int(n / avg * 1.15) + 64
    ids = np.searchsorted(cdf, rng.random(nwords)).clip(0, lens.size - 1).astype(np.int32)
    wl = lens[ids]
    out = np.full(int((wl + 1).sum()), 32, dtype=np.uint8)
    pos = np.concatenate([[0], np.cumsum(wl + 1)[:-1]])
    for j in range(int(lens.max())):
        m = np.flatnonzero(wl > j)
        out[pos[m] + j] = letters[starts[ids[m]] + j]
    idx = np.cumsum(rng.integers(8, 17, nwords // 8 + 2))
    idx = idx[idx < nwords]
    out[pos[idx] + wl[idx]] = 10
    return out[:n]


def d2_words_parts(n, seed=12345, vocab=50000, parts=16, threads=None):
    """D2 at sizes where d2_words (one generator, one thread: ~65 s per 1e9 characters) is too slow for a bench run: the same
    distribution -- Zipf(1.0) draws from ONE synthetic vocabulary, space separated, a newline every 8..16 words -- drawn as
    `parts` independent streams (default_rng([seed, k])) by a thread pool and concatenated.  A different text from
    d2_words(n, seed), the same generator per part."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    tabs = _d2_tables(seed, vocab)
    per = [n // parts + (1 if k < n % parts else 0) for k in range(parts)]
    workers = threads or max(1, min(parts, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(workers) as ex:
        outs = list(ex.map(lambda k: _d2_part(per[k], tabs, seed, k), range(parts)))
    return np.concatenate(outs)


def all_same(n, ch=97):
    return np.full(n, ch, dtype=np.uint8)


def periodic(n, period):
    base = np.arange(period, dtype=np.uint8) + 97
    return np.tile(base, n // period + 1)[:n].copy()


def fibonacci(n):
    a, b = b"a", b"ab"
    while len(b) < n:
        a, b = b, b + a
    return np.frombuffer(b[:n], dtype=np.uint8).copy()


def _splitmix64(x):
    """Vectorised splitmix64 finaliser (uint64 array in, uint64 array out)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def query_batch(text, q, m=16, seed=0, lo=0, hi=None):
    """Patterns lo..hi-1 (default: all) of the batch of Q = q patterns of m bytes (packed, offsets): even i = text
    window at pos = ((i * 0x9E3779B97F4A7C15) >> 11) % (N - m) with '\n' replaced by 'a'; odd i = uniform random
    lower-case string.  Every pattern is a function of (i, seed) alone, so a rank's slice of the global batch
    (bench.py --gpus N) is generated without the rest.  Returns (uint8[(hi-lo)*m], uint64[hi-lo+1])."""
    n = text.size
    hi = q if hi is None else hi
    cnt = hi - lo
    i = np.arange(lo, hi, dtype=np.uint64)
    with np.errstate(over="ignore"):
        pos = ((i * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(11)) % np.uint64(max(n - m, 1))
    pos = pos.astype(np.int64)
    win = text[pos[:, None] + np.arange(m)[None, :]] if n >= m else np.full((cnt, m), 97, np.uint8)
    win = np.where(win == 10, 97, win).astype(np.uint8)
    words = (m + 7) // 8
    with np.errstate(over="ignore"):
        h = _splitmix64(i * np.uint64(0xD1342543DE82EF95) + np.uint64((seed + 99) & 0xFFFFFFFF))
    cols = []
    for _ in range(words):
        cols.append(h.copy().view(np.uint8).reshape(cnt, 8))
        h = _splitmix64(h)
    rnd = (np.concatenate(cols, axis=1)[:, :m] % 26 + 97).astype(np.uint8) if cnt else np.zeros((0, m), np.uint8)
    odd = (np.arange(lo, hi) & 1).astype(bool)
    pats = np.where(odd[:, None], rnd, win).astype(np.uint8)
    off = (np.arange(cnt + 1, dtype=np.uint64) * np.uint64(m))
    return np.ascontiguousarray(pats.reshape(-1)), off
