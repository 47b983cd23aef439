"""Generates tests/golden/*.npz from the REFERENCE compiled in place (oracle/_ref, built by
oracle/Makefile from /root/reference: libsais 2.8.4 + engine.c).  Run in the authoring
container only:  python tests/golden/make_golden.py

The reference holds no golden vectors of its own for this path, so these are
outputs of the reference itself on inputs chosen per SURVEY.md 8(c):
  golden_small.npz   texts + libsais SA (int32) for classic / adversarial / boundary-length cases
  golden_readme.npz  README.md:16-20 three documents: text, libsais SA, get_substring_positions ranges
  golden_1mb.npz     1 MB slices of D1 / D2 / sigma=4: text, sha256 of libsais SA, 1000 query ranges
                     (get_substring_positions on the libsais SA, L = 32) and the same ranges on the
                     reference's own truncated SA (construct_truncated_suffix_array, L = 32)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Ref  # noqa: E402
from suffixarray_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PAIR = np.dtype([("first", "<u4"), ("second", "<u4")])


def small_cases():
    rng = np.random.default_rng(20241008)
    c = {
        "banana": b"banana", "mississippi": b"mississippi", "abracadabra": b"abracadabra",
        "all_a_1000": b"a" * 1000, "ab_500": b"ab" * 500, "abc_333": b"abc" * 333,
        "fib_f20": bytes(synth.fibonacci(10946)), "perm256_x3": bytes(range(256)) * 3,
        "highbit": bytes([255, 0, 128, 255, 255, 0, 0, 1, 254, 127]) * 40,
        "len1": b"x", "len2_ab": b"ab", "len2_ba": b"ba", "len2_aa": b"aa", "len3": b"cab",
    }
    for n in (63, 64, 65):
        c[f"rand27_{n}"] = rng.integers(97, 124, n, dtype=np.uint8).tobytes()
    for n in (65535, 65536, 65537):  # libsais switches code paths at 65536 (libsais.c:744)
        c[f"rand4_{n}"] = (rng.integers(0, 4, n, dtype=np.uint8) + 97).tobytes()
    c["rand256_4096"] = rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()
    c["rand2_20000"] = (rng.integers(0, 2, 20000, dtype=np.uint8) + 97).tobytes()
    return c


def main():
    ref = Ref()
    # ---- small ------------------------------------------------------------------------------
    arrays = {}
    names = []
    for name, text in small_cases().items():
        t = np.frombuffer(text, dtype=np.uint8)
        sa = ref.libsais(t, threads=1)
        sa_omp = ref.libsais(t, threads=4)
        sa64 = ref.libsais64(t, threads=1) if t.size > 1 else sa.astype(np.int64)
        assert np.array_equal(sa, sa_omp) and np.array_equal(sa.astype(np.int64), sa64)
        arrays[f"text__{name}"] = t
        arrays[f"sa__{name}"] = sa
        names.append(name)
    arrays["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "golden_small.npz"), **arrays)

    # ---- README documents (README.md:16-20) ---------------------------------------------------
    docs = ["The quick brown fox jumps over the lazy dog",
            "I am going to the store to buy some milk",
            "Uhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhh"]
    text = "\n".join(docs).lower().encode()
    t = np.frombuffer(text, dtype=np.uint8)
    sa = ref.libsais(t)
    tn = np.concatenate([t, np.zeros(64, np.uint8)])
    pats = [b"the quick brown fox", b"the", b"milk", b"zzz", b"uhhh", b"o", b"dog", b"to the store to buy some milk"]
    rr = np.zeros(len(pats), dtype=PAIR)
    for i, p in enumerate(pats):
        rr[i] = ref.query(tn, sa.astype(np.uint32), 32, p)
    np.savez_compressed(os.path.join(OUT, "golden_readme.npz"), text=t, sa=sa, patterns=np.array(pats),
                        ranges=rr, max_suffix_length=np.array([32]))

    # ---- 1 MB slices ----------------------------------------------------------------------------
    n = 1 << 20
    rng = np.random.default_rng(7)
    texts = {
        "d1": synth.d1_uniform27(n),
        "d2": synth.d2_words(n),
        "s4": np.where(rng.random(n) < 0.03, 10, rng.integers(97, 101, n)).astype(np.uint8),
    }
    arrays = {}
    L = 32
    for name, t in texts.items():
        sa = ref.libsais(t, threads=4)
        tn = np.concatenate([t, np.zeros(64, np.uint8)])
        tsa = ref.truncated_sa(tn, n, L)
        pats = []
        for i in range(1000):
            m = int(rng.integers(1, 41))
            if i % 2 == 0:
                p = int(rng.integers(0, n - m))
                q = bytes(t[p:p + m]).replace(b"\n", b"a")
            else:
                q = bytes(rng.integers(97, 123, m, dtype=np.uint8))
            pats.append(q)
        rr = np.zeros(len(pats), dtype=PAIR)
        rt = np.zeros(len(pats), dtype=PAIR)
        for i, p in enumerate(pats):
            rr[i] = ref.query(tn, sa.astype(np.uint32), L, p)
            rt[i] = ref.query(tn, tsa, L, p)
        arrays[f"text__{name}"] = t
        arrays[f"sa_sha256__{name}"] = np.array([hashlib.sha256(sa.astype("<i4").tobytes()).hexdigest()])
        arrays[f"patterns__{name}"] = np.array(pats)
        arrays[f"ranges__{name}"] = rr
        arrays[f"ranges_truncated_ref__{name}"] = rt
    arrays["names"] = np.array(list(texts))
    arrays["max_suffix_length"] = np.array([L])
    np.savez_compressed(os.path.join(OUT, "golden_1mb.npz"), **arrays)
    for f in ("golden_small.npz", "golden_readme.npz", "golden_1mb.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
