"""Shared input cases for the parity tests (same seeds on CPU and GPU)."""
import numpy as np
import pytest

from suffixarray_amd import synth


def small_texts():
    rng = np.random.default_rng(42)
    c = {
        "banana": np.frombuffer(b"banana", np.uint8),
        "mississippi": np.frombuffer(b"mississippi", np.uint8),
        "len1": np.frombuffer(b"z", np.uint8),
        "len2": np.frombuffer(b"ba", np.uint8),
        "aa": np.frombuffer(b"aa", np.uint8),
        "all_a_5000": synth.all_same(5000),
        "all_a_70000": synth.all_same(70000),
        "ab_3000": synth.periodic(6000, 2),
        "abc_2000": synth.periodic(6000, 3),
        "period7": synth.periodic(50001, 7),
        "fib": synth.fibonacci(46368),
        "perm256": np.tile(np.arange(256, dtype=np.uint8), 20),
        "highbit": np.tile(np.frombuffer(bytes([255, 0, 128, 255, 255, 0, 0, 1]), np.uint8), 300),
        "zeros": np.zeros(3000, np.uint8),
        "with_nul": np.frombuffer(b"ab\x00ab\x00\x00abab\x00", np.uint8),
    }
    for n in (3, 63, 64, 65, 4095, 4096, 4097, 8193, 65535, 65536, 65537):
        c[f"r27_{n}"] = rng.integers(97, 124, n, dtype=np.uint8)
    for sig in (2, 4, 256):
        c[f"r{sig}_30000"] = rng.integers(0, sig, 30000, dtype=np.uint8)
    c["d1_300k"] = synth.d1_uniform27(300_000)
    c["d2_300k"] = synth.d2_words(300_000)
    # long repeats: a random block repeated, forces doubling rounds on a large active set
    blk = rng.integers(97, 101, 5000, dtype=np.uint8)
    c["repeat_block"] = np.tile(blk, 12)
    return c


def query_patterns(text, count, rng, maxlen=40):
    n = text.size
    pats = []
    for i in range(count):
        m = int(rng.integers(1, maxlen + 1))
        if i % 2 == 0 and n > m:
            p = int(rng.integers(0, n - m))
            pats.append(bytes(text[p:p + m]))
        else:
            pats.append(bytes(rng.integers(97, 123, m, dtype=np.uint8)))
    pats += [b"", b"a", b"zzzzzzzz", bytes([255]) * 3, bytes([1]), bytes(text[-5:]), bytes(text[-1:]), bytes(text[:7])]
    return pats


def check_partitioned(SuffixArray, tmp_path):
    """documents cut into partitions of whole documents (the reference's scheme for inputs beyond one index, pyx:148-180,
    221-247): the same records as one index over everything, k honoured across partitions, save / load."""
    rng = np.random.default_rng(17)
    words = ["alpha", "beta", "gamma", "delta", "Milk", "store", "fox", "lazy dog", "quick", "brown"]
    docs = [" ".join(words[j] for j in rng.integers(0, len(words), rng.integers(2, 9))) + (" #%d" % i) for i in range(120)]
    one = SuffixArray(documents=docs, max_suffix_length=32)
    part = SuffixArray(documents=docs, max_suffix_length=32, partition_bytes=700)
    assert len(part.partitions) >= 5 and len(one.partitions) == 1
    for q in ("milk", "lazy dog", "FOX", "#7", "zzz", "a", "alpha beta"):
        exp = [d for d in docs if q.lower() in d.lower()]
        assert sorted(one.query_records(q, k=10**6)) == sorted(exp)
        assert sorted(part.query_records(q, k=10**6)) == sorted(exp), q
        few = part.query_records(q, k=3)
        assert len(few) == min(3, len(exp)) and all(r in exp for r in few)
    got = part.query_records_batch(["milk", "", "delta", "zzz"], k=5)
    assert got[1] == [] and got[3] == [] and len(got[0]) == min(5, sum("milk" in d.lower() for d in docs)) and all("delta" in r.lower() for r in got[2])
    with pytest.raises(RuntimeError):
        part.query_ranges(["milk"])
    part.save(str(tmp_path / "parts"))
    back = SuffixArray.load(str(tmp_path / "parts"))
    assert len(back.partitions) == len(part.partitions)
    assert sorted(back.query_records("quick", k=10**6)) == sorted(d for d in docs if "quick" in d.lower())
    for x in (one, part, back):
        x.close()


def check_partitioned_csv(SuffixArray, tmp_path):
    """a CSV column cut into partitions of whole rows (sa_hip_csv_index_create_partitioned; the reference cuts the file every
    2 GiB, engine.c:1437-1481, and answers from the partitions one after the other, pyx:221-247): the same rows as one index over
    the whole column, quoted fields and commas inside them included, k honoured across partitions, save / load."""
    import csv as _csv
    rng = np.random.default_rng(23)
    words = ["Acme", "Globex", "Initech", "Umbrella", "Hooli", "Vehement", "Massive Dynamic", "Stark", "Wayne", "Wonka"]
    tails = ["", " Inc", " LLC", ", Inc.", " Ltd", ' "The Best"']
    rows = [(str(i), words[int(rng.integers(0, len(words)))] + " " + words[int(rng.integers(0, len(words)))] + tails[int(rng.integers(0, len(tails)))],
             ["US", "DE", "FR"][i % 3]) for i in range(600)]
    path = str(tmp_path / "companies.csv")
    with open(path, "w", newline="") as f:
        w = _csv.writer(f)
        w.writerow(["id", "company_name", "country"])
        w.writerows(rows)
    one = SuffixArray(csv_file=path, search_column="company_name", max_suffix_length=32)
    part = SuffixArray(csv_file=path, search_column="company_name", max_suffix_length=32, partition_bytes=1500)
    assert len(one.partitions) == 1 and len(part.partitions) >= 6
    assert part.columns == ["id", "company_name", "country"]

    def ids(recs):
        return sorted(int(r["id"]) for r in recs)
    for q in ("acme", "INC", ", inc.", "massive dynamic", "the best", "zzz", "a", "hooli w"):
        exp = sorted(int(r[0]) for r in rows if q.lower() in r[1].lower())
        assert ids(one.query_records(q, k=10**6)) == exp, q
        got = part.query_records(q, k=10**6)
        assert ids(got) == exp, q
        assert all(set(r) == {"id", "company_name", "country"} for r in got)
        few = part.query_records(q, k=4)
        assert len(few) == min(4, len(exp)) and all(int(r["id"]) in exp for r in few)
    got = part.query_records_batch(["acme", "", "wonka", "zzz"], k=7)
    assert got[1] == [] and got[3] == [] and len(got[0]) == min(7, sum("acme" in r[1].lower() for r in rows))
    assert all("wonka" in r["company_name"].lower() for r in got[2])
    part.save(str(tmp_path / "csvparts"))
    back = SuffixArray.load(str(tmp_path / "csvparts"))
    assert len(back.partitions) == len(part.partitions) and back.columns == part.columns
    assert ids(back.query_records("stark", k=10**6)) == sorted(int(r[0]) for r in rows if "stark" in r[1].lower())
    for x in (one, part, back):
        x.close()
