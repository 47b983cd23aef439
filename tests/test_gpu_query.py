"""GPU: batched substring query through the C ABI against the oracle restatement of
get_substring_positions (engine.c:869-918) and the reference-generated golden ranges."""
import os

import numpy as np
import pytest

import cases
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_golden_readme(gpu):
    g = np.load(os.path.join(GOLDEN, "golden_readme.npz"), allow_pickle=False)
    t, sa = g["text"], g["sa"].astype(np.uint32)
    L = int(g["max_suffix_length"][0])
    pats = [bytes(p) for p in g["patterns"]]
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.load(t, sa, L)
        assert np.array_equal(idx.query_batch(pats), g["ranges"])
        idx.build(t, L)   # truncated device build gives the same ranges
        assert np.array_equal(idx.query_batch(pats), g["ranges"])
        idx.build(t, 0)   # and so does the full suffix array
        assert np.array_equal(idx.query_batch(pats), g["ranges"])


def test_golden_1mb_ranges(gpu):
    g = np.load(os.path.join(GOLDEN, "golden_1mb.npz"), allow_pickle=False)
    L = int(g["max_suffix_length"][0])
    with gpu.DeviceIndex(1 << 20, 0) as idx:
        for name in g["names"]:
            t = g[f"text__{name}"]
            pats = [bytes(p) for p in g[f"patterns__{name}"]]
            idx.build(t, L)
            got = idx.query_batch(pats)
            assert np.array_equal(got, g[f"ranges__{name}"]), name
            # hit sets, not only ranges
            sa = idx.sa_u32()
            for q, (f, s) in list(zip(pats, got))[:200]:
                if f != 0xFFFFFFFF and s != 0xFFFFFFFF and s >= f:
                    for p in sa[f:s + 1][:20]:
                        assert bytes(t[p:p + min(len(q), L)]) == q[:L]


def test_small_cases_vs_oracle(gpu, oracle):
    rng = np.random.default_rng(123)
    texts = cases.small_texts()
    nmax = max(t.size for t in texts.values())
    with gpu.DeviceIndex(nmax, 0) as idx:
        for name, t in texts.items():
            sa = oracle.sais(t).astype(np.uint32)
            pats = cases.query_patterns(t, 300, rng)
            for L in (0, 4, 32):
                idx.load(t, sa, L)
                got = idx.query_batch(pats)
                exp = oracle.query_batch(t, sa, L if L else 0xFFFFFFFF, pats)
                assert np.array_equal(got, exp), (name, L)


def test_empty_batch_and_empty_text(gpu):
    with gpu.DeviceIndex(64, 0) as idx:
        idx.build(np.frombuffer(b"hello\nworld\n", np.uint8))
        assert idx.query_batch([]).size == 0
        r = idx.query_batch([b""])
        assert tuple(r[0]) == (0, 11)
        idx.build(np.zeros(0, np.uint8))
        r = idx.query_batch([b"a", b""])
        assert tuple(r[0]) == (0xFFFFFFFF, 0xFFFFFFFF) and tuple(r[1]) == (0xFFFFFFFF, 0xFFFFFFFF)


def test_single_query_wrapper(gpu, oracle):
    t = np.frombuffer(b"the quick brown fox\njumps over the lazy dog\n", np.uint8)
    sa = oracle.sais(t).astype(np.uint32)
    for q in (b"the", b"fox", b"zzz", b"q", b"the lazy dog\n"):
        assert gpu.get_substring_positions(t, sa, 32, q) == oracle.query(t, sa, 32, q), q


def test_large_batch_d1(gpu, oracle):
    from suffixarray_amd import synth
    t = synth.d1_uniform27(5_000_000)
    buf, off = synth.query_batch(t, 200_000, 16)
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.build(t)
        got = idx.query_batch((buf, off))
        sa = idx.sa_u32()
    exp = oracle.query_batch(t, sa, 0xFFFFFFFF, (buf, off))
    assert np.array_equal(got, exp)
    hits = ((got["second"] - got["first"] + 1) & 0xFFFFFFFF) > 0
    assert 0.05 < hits.mean() < 0.95


def test_python_api_documents(gpu):
    from suffixarray_amd import SuffixArray
    docs = ["The quick brown fox jumps over the lazy dog",
            "I am going to the store to buy some milk",
            "Uhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhh"]
    s = SuffixArray(documents=docs, max_suffix_length=32)
    assert s.query_records("the quick brown fox") == [docs[0]]
    assert sorted(s.query_records("the")) == sorted(docs[:2])
    assert s.query_records("MILK") == [docs[1]]
    assert s.query_records("zzz") == []
    assert s.query_records("") == []
    assert s.query_records("the", k=1) in ([docs[0]], [docs[1]])
    assert s.query_records_batch(["milk", "", "uhh"]) == [[docs[1]], [], [docs[2]]]
    s.close()


def test_python_api_partitioned_csv(gpu, tmp_path):
    from suffixarray_amd import SuffixArray
    cases.check_partitioned_csv(SuffixArray, tmp_path)


def test_python_api_partitioned_documents(gpu, tmp_path):
    from suffixarray_amd import SuffixArray
    cases.check_partitioned(SuffixArray, tmp_path)


def test_python_api_csv(gpu, tmp_path):
    from suffixarray_amd import SuffixArray
    p = tmp_path / "companies.csv"
    p.write_text('id,company_name,country\n1,Netflix,US\n2,"Acme, Inc.",US\n3,Initech,DE\n4,netflix studios,US\n')
    s = SuffixArray(csv_file=str(p), search_column="company_name", max_suffix_length=32)
    recs = s.query_records("netflix")
    assert sorted(r["id"] for r in recs) == ["1", "4"]
    assert s.query_records("acme, inc")[0] == {"id": "2", "company_name": "Acme, Inc.", "country": "US"}
    assert s.query_records("company_name") == []   # header row is not indexed
    s.close()


def test_one_python_class_on_the_c_seam(gpu, tmp_path):
    """suffixarray_amd.SuffixArray IS the Cython class (one implementation; it binds the C seam with `cdef extern`)."""
    from suffixarray_amd.suffix_array import SuffixArray as CySuffixArray
    from suffixarray_amd import SuffixArray
    assert SuffixArray is CySuffixArray
    docs = ["The quick brown fox jumps over the lazy dog",
            "I am going to the store to buy some milk",
            "Uhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhh"]
    a = SuffixArray(documents=docs, max_suffix_length=32)
    assert a.query_records("the quick brown fox") == [docs[0]]
    r = a.query_ranges(["the", "fox", "qqq", "zzz"])
    assert ((r["second"] - r["first"] + 1) & 0xFFFFFFFF).tolist()[:3] == [3, 1, 0]
    assert (int(r["first"][3]), int(r["second"][3])) == (0xFFFFFFFF, 0xFFFFFFFF)   # every suffix is smaller (engine.c:896-898)
    assert a._index.verify() == 0 and a._index.n == len("\n".join(docs))
    p = tmp_path / "c.csv"
    p.write_text('id,company_name,country\n1,Netflix,US\n2,"Acme, Inc.",US\n3,netflix studios,US\n')
    c = SuffixArray(csv_file=str(p), search_column="company_name", max_suffix_length=32)
    assert sorted(r["id"] for r in c.query_records("netflix")) == ["1", "3"]
    assert c.query_records("netflix", k=1)[0]["id"] in ("1", "3") and len(c.query_records("netflix", k=1)) == 1
    assert c.columns == ["id", "company_name", "country"]
    a.close(); c.close()


def test_csv_mode_against_the_reference(gpu, tmp_path):
    """CSV mode pinned to the reference's own C path (tests/golden/golden_csv.npz, generated by make_golden.py from
    construct_truncated_suffix_array_from_csv_partitioned_mmap + get_substring_positions_file +
    get_matching_records_file of the compiled reference): a tiny file with a quoted field and 10 000 rows of the
    synthetic config-5 generator.  For every pattern without a byte below ',' the C seam's hit count and the row
    set of query_records equal the reference's; patterns with a space are compared where the reference agrees with
    a plain scan of the column (its binary search over FILE bytes is not monotone for them, see make_golden.py) and
    against the scan itself everywhere."""
    import csv as _csv
    import io
    from suffixarray_amd import SuffixArray
    g = np.load(os.path.join(GOLDEN, "golden_csv.npz"), allow_pickle=False)
    L = int(g["max_suffix_length"][0])
    for name in g["names"]:
        data = bytes(g[f"csv__{name}"])
        column = str(g[f"column__{name}"][0])
        path = tmp_path / f"{name}.csv"
        path.write_bytes(data)
        rows = list(_csv.reader(io.StringIO(data.decode())))
        ci = rows[0].index(column)
        fields = [(int(r[0]), r[ci].lower()) for r in rows[1:]]
        pats, off, ids = g[f"patterns__{name}"], g[f"row_ids_offsets__{name}"], g[f"row_ids__{name}"]
        s = SuffixArray(csv_file=str(path), search_column=column, max_suffix_length=L)
        checked = 0
        with gpu.CsvIndex(str(path), column, L) as c:
            assert c.num_rows == len(fields) and c.columns == rows[0]
            # the reference indexes the header row too (n__ counts its characters); this index does not
            assert c.index.n == int(g[f"n__{name}"][0]) - (len(column) + 1)
            for i, p in enumerate(pats):
                p = str(p)
                ref_ids = ids[off[i]:off[i + 1]].tolist()
                scan = sorted(j for j, f in fields if p in f)
                got = sorted(int(r["id"]) for r in s.query_records(p.upper(), k=len(fields)))
                assert got == scan, (name, p)
                first, second = c.get_substring_positions_file(p.encode())
                assert first != 0xFFFFFFFF
                hits = sum(f.count(p) if len(p) == 1 else sum(1 for o in range(len(f)) if f.startswith(p, o)) for _, f in fields)
                assert second - first + 1 == hits, (name, p)
                if bool(g[f"letters_only__{name}"][i]) or bool(g[f"agrees_with_scan__{name}"][i]):
                    assert got == ref_ids, (name, p)
                if bool(g[f"letters_only__{name}"][i]):
                    assert second - first + 1 == int(g[f"hit_counts__{name}"][i]), (name, p)
                    checked += 1
                raw, num = c.get_matching_records_file(p.encode(), len(fields))
                assert num == len(raw) == len(scan)
                assert sorted(int(r.split(b",", 1)[0]) for r in raw) == scan
            # any miss -> {UINT32_MAX, UINT32_MAX} (engine.c:962-965); nothing is appended
            for miss in (b"zzzzqq", b"company_name", b"\xff"):
                assert c.get_substring_positions_file(miss) == (0xFFFFFFFF, 0xFFFFFFFF)
                assert c.get_matching_records_file(miss, 10) == ([], 0)
        assert checked >= (5 if name == "tiny" else 100)
        s.close()


def test_record_retrieval_seam_conventions(gpu, oracle, tmp_path):
    """get_matching_records_file's calling convention (engine.c:1326-1390): rows are appended from *num_matches on, at
    most k in the table overall, every row a malloc'ed string the caller frees; whole rows, each once.  And the
    in-memory form sa_hip_get_matching_records (engine.c:1168-1215) on a documents text."""
    p = tmp_path / "c.csv"
    p.write_bytes(b'id,company_name,country\r\n1,Netflix,US\r\n2,"Acme, Inc.",US\r\n3,netflix netflix studios,US\r\n4,Netflix BV,NL')
    with gpu.CsvIndex(str(p), "company_name", 32) as c:
        rows, num = c.get_matching_records_file(b"netflix", 10)
        assert num == 3 and sorted(rows) == [b"1,Netflix,US", b"3,netflix netflix studios,US", b"4,Netflix BV,NL"]   # row 3 once, no CR / LF
        rows, num = c.get_matching_records_file(b"netflix", 2)
        assert num == 2 and len(rows) == 2
        rows, num = c.get_matching_records_file(b"netflix", 3, already=(b"x", b"y"))   # two slots taken: one more fits
        assert num == 3 and len(rows) == 1
        rows, num = c.get_matching_records_file(b"netflix", 2, already=(b"x", b"y"))   # table full
        assert num == 2 and rows == []
        assert c.get_matching_records_file(b"acme, inc", 5)[0] == [b'2,"Acme, Inc.",US']
    docs = [b"the quick brown fox", b"jumps over the lazy dog", b"milk", b"the end"]
    text = np.frombuffer(b"\n".join(docs), np.uint8)
    sa = oracle.sais(text).astype(np.uint32)
    assert sorted(gpu.get_matching_records(text, sa, 32, b"the", 10)) == sorted([docs[0], docs[1], docs[3]])
    assert gpu.get_matching_records(text, sa, 32, b"the", 1)[0] in (docs[0], docs[1], docs[3])
    assert gpu.get_matching_records(text, sa, 32, b"milk", 4) == [b"milk"]
    assert gpu.get_matching_records(text, sa, 32, b"zzz", 4) == []


def test_config5_csv_mode_reduced_scale(gpu, tmp_path):
    """BASELINE config 5 at reduced scale: synthetic company_name CSV, max_suffix_length = 32,
    query_records against a brute-force scan of the column."""
    from suffixarray_amd import SuffixArray
    from csv_ingest import extract_column
    path = tmp_path / "companies.csv"
    gpu.synth_csv(str(path), 200_000, 11)
    col = extract_column(str(path), "company_name")
    names = col.text.split(b"\n")[:-1]
    s = SuffixArray(csv_file=str(path), search_column="company_name", max_suffix_length=32)
    rng = np.random.default_rng(4)
    queries = [names[i].decode() for i in rng.integers(0, len(names), 60)]
    queries += [names[i].decode()[1:7] for i in rng.integers(0, len(names), 20)] + ["zzzzqqqq", ", inc."]
    got = s.query_records_batch([q.upper() for q in queries], k=1000)   # case-insensitive like the reference
    for q, recs in zip(queries, got):
        ql = q.lower().encode()[:32]
        exp = {i + 1 for i, nm in enumerate(names) if ql in nm}
        ids = {int(r["id"]) for r in recs}
        if len(exp) <= 1000:
            assert ids == exp, q
        else:
            assert len(ids) == 1000 and ids <= exp, q
        for r in recs[:5]:
            assert ql in r["company_name"].lower().encode()
    s.close()


def test_save_load_roundtrip(gpu, tmp_path):
    from suffixarray_amd import SuffixArray
    docs = ["The quick brown fox jumps over the lazy dog", "I am going to the store to buy some milk", "Uhhhhhhhh"]
    a = SuffixArray(documents=docs, max_suffix_length=32)
    a.save(str(tmp_path / "idx_docs"))
    b = SuffixArray.load(str(tmp_path / "idx_docs"))
    for q in ("the", "milk", "uhh", "zzz", "fox jumps"):
        assert a.query_records(q) == b.query_records(q)
    p = tmp_path / "c.csv"
    p.write_text('id,company_name,country\n1,Netflix,US\n2,"Acme, Inc.",US\n3,netflix studios,US\n')
    c = SuffixArray(csv_file=str(p), search_column="company_name", max_suffix_length=32)
    c.save(str(tmp_path / "idx_csv"))
    d = SuffixArray.load(str(tmp_path / "idx_csv"))
    assert c.query_records("netflix") == d.query_records("netflix") and len(d.query_records("netflix")) == 2
    assert d.query_records("acme, inc")[0]["id"] == "2"
    assert len(d.query_records("netflix", k=10**9)) == 2        # k = "all" sizes nothing by k
    for x in (a, b, c, d):
        x.close()


def _saved_arrays(directory):
    return (np.fromfile(os.path.join(directory, "text.u8"), dtype=np.uint8), np.fromfile(os.path.join(directory, "sa.u32"), dtype=np.uint32))


def test_loaded_documents_index_against_the_oracle(gpu, oracle, tmp_path):
    """after load() nothing is compared with the index it came from -- the loaded index answers a pattern
    batch exactly as the oracle does on the SAVED text and suffix array, the saved array is the oracle's truncated
    suffix array of the saved text, and records come back as a plain scan of the documents finds them (>= 1e6 characters)."""
    from suffixarray_amd import SuffixArray, synth
    text = synth.d2_words(1_300_000)
    docs = [d.title() for d in bytes(text).decode().split("\n")]          # mixed case: the index lower-cases
    L = 24
    a = SuffixArray(documents=docs, max_suffix_length=L)
    a.save(str(tmp_path / "idx"))
    a.close()
    b = SuffixArray.load(str(tmp_path / "idx"))
    t, sa = _saved_arrays(str(tmp_path / "idx"))
    assert t.size >= 1_000_000 and bytes(t) == "\n".join(docs).lower().encode()
    assert np.array_equal(sa, oracle.truncated_sa(t, L))
    rng = np.random.default_rng(8)
    pats = cases.query_patterns(t, 3000, rng)
    pats = [p for p in pats if p == p.lower() and b"\0" not in p]         # query_ranges lower-cases what it is given
    got = b.query_ranges([bytes(p) for p in pats])
    exp = oracle.query_batch(t, sa, L, pats)
    assert np.array_equal(got["first"], exp["first"]) and np.array_equal(got["second"], exp["second"])
    assert b._index.verify() == 0
    low = [d.lower() for d in docs]
    for q in ["the", docs[17][:9], docs[-1][2:12].upper(), "zzzzzq", docs[5]]:
        exp_rows = [d for d, l in zip(docs, low) if q.lower()[:L] in l]
        got_rows = b.query_records(q, k=10**9)
        assert sorted(got_rows) == sorted(exp_rows), q
    b.close()


def test_loaded_csv_index_against_the_oracle_and_refusals(gpu, oracle, tmp_path):
    """The same for a CSV-mode index of 120 000 rows; then what load() must refuse: a truncated array file, a row table
    that does not ascend, a suffix array with an entry >= n, a CSV file that changed after the index was saved."""
    import json
    import shutil
    from suffixarray_amd import SuffixArray
    from csv_ingest import extract_column
    path = tmp_path / "companies.csv"
    gpu.synth_csv(str(path), 120_000, 3)
    L = 32
    a = SuffixArray(csv_file=str(path), search_column="company_name", max_suffix_length=L)
    good = str(tmp_path / "idx")
    a.save(good)
    a.close()
    b = SuffixArray.load(good)
    t, sa = _saved_arrays(good)
    col = extract_column(str(path), "company_name")
    assert bytes(t) == bytes(col.text)
    assert np.array_equal(sa, oracle.truncated_sa(t, L))
    names = bytes(t).split(b"\n")[:-1]
    rng = np.random.default_rng(2)
    pats = [names[i] for i in rng.integers(0, len(names), 1500)] + [names[i][1:8] for i in rng.integers(0, len(names), 500)] + [b"zzqq", b", inc."]
    got = b.query_ranges([p.decode() for p in pats])
    exp = oracle.query_batch(t, sa, L, pats)
    assert np.array_equal(got["first"], exp["first"]) and np.array_equal(got["second"], exp["second"])
    for q in [names[5].decode(), names[77].decode()[:5], "llc"]:
        ids = {int(r["id"]) for r in b.query_records(q, k=10**9)}
        assert ids == {i + 1 for i, nm in enumerate(names) if q.encode()[:L] in nm}, q
    b.close()

    def variant(name, mutate):
        d = str(tmp_path / name)
        shutil.copytree(good, d)
        mutate(d)
        return d

    def truncate(d):
        with open(os.path.join(d, "sa.u32"), "r+b") as f:
            f.truncate(os.path.getsize(os.path.join(d, "sa.u32")) - 4096)

    def scramble_rows(d):
        r = np.fromfile(os.path.join(d, "row_file_offsets.u64"), dtype=np.uint64)
        r[1000], r[2000] = r[2000], r[1000]
        r.tofile(os.path.join(d, "row_file_offsets.u64"))

    def bad_entry(d):
        s = np.fromfile(os.path.join(d, "sa.u32"), dtype=np.uint32)
        s[12345] = s.size + 5
        s.tofile(os.path.join(d, "sa.u32"))

    with pytest.raises(ValueError, match="truncated"):
        SuffixArray.load(variant("trunc", truncate))
    with pytest.raises((RuntimeError, ValueError), match="ascend"):
        SuffixArray.load(variant("rows", scramble_rows))
    with pytest.raises(RuntimeError, match=">= n"):
        SuffixArray.load(variant("entry", bad_entry))
    # the CSV file changes after the index was saved: same size, other modification time; then another size
    st = os.stat(path)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000_000))
    with pytest.raises(ValueError, match="changed"):
        SuffixArray.load(good)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns))
    SuffixArray.load(good).close()
    with open(path, "ab") as f:
        f.write(b"120001,late arrival llc,US\n")
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns))
    with pytest.raises(ValueError, match="changed"):
        SuffixArray.load(good)
    meta = json.load(open(os.path.join(good, "meta.json")))
    assert meta["version"] == 3 and meta["csv_size"] == st.st_size


def test_device_rows_equal_host_rows(gpu, monkeypatch):
    """Record retrieval on the device (rows_device.hpp: one workgroup per query, hits -> rows by binary search over the
    row table, per-query de-duplication in an LDS hash table, rows in SA order of their first hit) against the host
    path it replaces (records.hpp: distinct_rows over copied SA slabs) -- the same rows in the same order for every k up to
    4096 (both table sizes), single query and batch, ranges of a few million hits in a handful of rows included -- and
    both against a plain scan of the column."""
    from csv_ingest import extract_column
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "c.csv")
        gpu.synth_csv(path, 300_000, 11)
        col = extract_column(path, "company_name")
    text = np.frombuffer(col.text, dtype=np.uint8)
    names = bytes(col.text).split(b"\n")[:-1]
    starts = np.asarray(col.text_row_starts, dtype=np.uint64)
    rng = np.random.default_rng(6)
    pats = [names[i] for i in rng.integers(0, len(names), 150)] + [names[i][2:6] for i in rng.integers(0, len(names), 60)]
    pats += [b"inc", b" ", b"a", b"llc", b"zzqqzz", b"e", names[0], b", inc.", b"\n"]
    with gpu.DeviceIndex(text.size, 0) as idx:
        idx.build(text, 32)
        idx.set_rows(starts)
        sa = idx.sa_u32()
        for k in (1, 7, 300, 1000, 1536, 1537, 4096, 5000):
            monkeypatch.setenv("SA_HIP_HOST_ROWS", "1")
            host_rows, host_rg = idx.query_rows_batch(pats, k)
            monkeypatch.delenv("SA_HIP_HOST_ROWS")
            dev_rows, dev_rg = idx.query_rows_batch(pats, k)
            assert np.array_equal(host_rg, dev_rg)
            for p, a, b in zip(pats, host_rows, dev_rows):
                assert np.array_equal(a, b), (k, p, a[:10], b[:10])
            for p, b in list(zip(pats, dev_rows))[::7]:
                one, rg1 = idx.query_rows(p, k)
                assert np.array_equal(one, b), (k, p)
        # k = 1000 against a scan: first k distinct rows in hit order
        for p, rows, rg in zip(pats, dev_rows if False else idx.query_rows_batch(pats, 1000)[0], dev_rg):
            exp = {i for i, nm in enumerate(names) if p[:32] in nm + b"\n"} if p != b"\n" else set(range(len(names)))
            got = set(int(r) for r in rows)
            if len(exp) <= 1000:
                assert got == exp, p
            else:
                assert len(got) == 1000 and got <= exp, p
            f, s2 = int(rg["first"]), int(rg["second"])
            if len(exp):
                hit_rows = np.searchsorted(starts, sa[f:min(s2 + 1, f + 200000)], side="right") - 1
                _, first_idx = np.unique(hit_rows, return_index=True)
                order = hit_rows[np.sort(first_idx)][:1000]
                assert np.array_equal(order[:rows.size], rows[:order.size]), p
    with gpu.DeviceIndex(100, 0) as idx:
        idx.build(np.frombuffer(b"abc\nabd\n", np.uint8))
        with pytest.raises(gpu.SaHipError):
            idx.query_rows_batch([b"ab"], 5)            # no row table
        idx.set_rows(np.array([0, 4], dtype=np.uint64))
        rows, rg = idx.query_rows_batch([b"ab", b"d", b"zz", b""], 10**9)
        assert [r.tolist() for r in rows] == [[0, 1], [1], [], [0, 1]] or [sorted(r.tolist()) for r in rows] == [[0, 1], [1], [], [0, 1]]


def test_large_rows_batch_through_the_pinned_ring(gpu, monkeypatch):
    """sa_hip_index_query_rows_batch on a batch whose Q x k row ids exceed 32 MiB: the host legs go through the ring of pinned
    slabs (host_io.hpp: ring_upload / ring_download / ring_download_pieces -- worker threads widen the u32 ids of the pieces that
    have arrived into the caller's uint64[Q][k]) and must hand back exactly what the plain copies (SA_HIP_ROWS_RING=0) do:
    counts, ranges and every live row id; Q and k chosen so that pieces end inside the batch and k * 4 does not divide a slab.
    A sample of the queries is checked against the one-query call (same contract as engine.c:1326-1390 per element)."""
    from csv_ingest import extract_column
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "c.csv")
        gpu.synth_csv(path, 200_000, 12)
        col = extract_column(path, "company_name")
    text = np.frombuffer(col.text, dtype=np.uint8)
    starts = np.asarray(col.text_row_starts, dtype=np.uint64)
    rng = np.random.default_rng(8)
    Q, k = 2_000_003, 13
    pos = rng.integers(0, text.size - 12, Q).astype(np.int64)
    lens = rng.integers(1, 9, Q).astype(np.int64)          # short substrings of the column: most hit several rows, some cross a row end
    off = np.zeros(Q + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    idxs = np.repeat(pos - off[:-1].astype(np.int64), lens) + np.arange(int(off[-1]), dtype=np.int64)
    buf = np.ascontiguousarray(text[idxs])
    miss = rng.random(Q) < 0.2                              # a fifth of the patterns get a byte that never occurs: misses
    buf[off[:-1][miss].astype(np.int64)] = 1
    with gpu.DeviceIndex(text.size, 0) as idx:
        idx.build(text, 32)
        idx.set_rows(starts)
        monkeypatch.setenv("SA_HIP_ROWS_RING", "0")
        (rows0, cnt0), rg0 = idx.query_rows_batch_raw((buf, off), k)
        monkeypatch.delenv("SA_HIP_ROWS_RING")
        (rows1, cnt1), rg1 = idx.query_rows_batch_raw((buf, off), k)
        (rows2, cnt2), rg2 = idx.query_rows_batch_raw((buf, off), k)      # the ring a second time (slabs and events reused)
        assert np.array_equal(cnt0, cnt1) and np.array_equal(cnt0, cnt2)
        assert np.array_equal(rg0, rg1) and np.array_equal(rg0, rg2)
        live = np.arange(k)[None, :] < cnt0[:, None]
        assert np.array_equal(rows0[live], rows1[live]) and np.array_equal(rows0[live], rows2[live])
        assert int(cnt0.max()) == k and 0.1 < float((cnt0 == 0).mean()) < 0.5
        # ranges of at most 4 hits (misses included) are answered by one lane each, the rest by a workgroup per query
        # (rows_device.hpp: rows_lane_kernel + the pending list): against every query through the workgroup form ...
        monkeypatch.setenv("SA_HIP_ROWS_LANES", "0")
        (rows3, cnt3), rg3 = idx.query_rows_batch_raw((buf, off), k)
        monkeypatch.delenv("SA_HIP_ROWS_LANES")
        assert np.array_equal(cnt0, cnt3) and np.array_equal(rg0, rg3) and np.array_equal(rows0[live], rows3[live])
        # ... ranges of 5 .. 4096 hits by one WAVE each for k <= 64 (rows_wave_kernel): against lanes + workgroups only
        monkeypatch.setenv("SA_HIP_ROWS_WAVES", "0")
        (rows4, cnt4), rg4 = idx.query_rows_batch_raw((buf, off), k)
        monkeypatch.delenv("SA_HIP_ROWS_WAVES")
        assert np.array_equal(cnt0, cnt4) and np.array_equal(rg0, rg4) and np.array_equal(rows0[live], rows4[live])
        hits = ((rg0["second"].astype(np.int64) - rg0["first"].astype(np.int64) + 1) & 0xFFFFFFFF) * (rg0["first"] != 0xFFFFFFFF)
        assert (hits == 0).sum() > 1000 and ((hits >= 1) & (hits <= 4)).sum() > 1000 and ((hits > 4) & (hits <= 4096)).sum() > 1000 \
            and (hits > 4096).sum() > 1000, np.bincount(np.minimum(hits, 6))
        # ... and a slice of the batch against the host path (records.hpp: distinct_rows), for several k
        sl = slice(1_000_000, 1_030_000)
        sub = (buf[int(off[sl.start]):int(off[sl.stop])], (off[sl.start:sl.stop + 1] - off[sl.start]).astype(np.uint64))
        for kk in (1, 2, 3, 13, 40, 64, 65, 100):
            monkeypatch.setenv("SA_HIP_HOST_ROWS", "1")
            (hr, hc), hrg = idx.query_rows_batch_raw(sub, kk)
            monkeypatch.delenv("SA_HIP_HOST_ROWS")
            (dr, dc), drg = idx.query_rows_batch_raw(sub, kk)
            lv = np.arange(kk)[None, :] < hc[:, None]
            assert np.array_equal(hc, dc) and np.array_equal(hrg, drg) and np.array_equal(hr[lv], dr[lv]), kk
            if kk == k:
                assert np.array_equal(dc, cnt1[sl]) and np.array_equal(dr[lv], rows1[sl][lv])
        for q in rng.integers(0, Q, 40):
            p = bytes(buf[int(off[q]):int(off[q + 1])])
            one, rg = idx.query_rows(p, k)
            assert np.array_equal(one, rows1[q, :cnt1[q]]), (q, p)
            assert (int(rg[0]), int(rg[1])) == (int(rg1[q]["first"]), int(rg1[q]["second"])), (q, p)


def test_rows_batch_wave_form_hands_long_ranges_on(gpu, monkeypatch):
    """Many hits in few rows: a range whose first 4096 hits do not yield k distinct rows is given up by the wave form and
    walked again by the workgroup form (rows_device.hpp: handoff list) -- three long rows of one letter, a batch large enough
    for the lane / wave forms, against the host path and against the batch without the wave form."""
    row = b"a" * 20_000 + b"b" * 50 + b"\n"
    text = np.frombuffer(row * 3 + b"abab\nba\n", np.uint8)
    starts = np.array([0, len(row), 2 * len(row), 3 * len(row), 3 * len(row) + 5], dtype=np.uint64)
    base = [b"a", b"aa", b"aaaaaaaa", b"b", b"bb", b"ab", b"ba", b"a" * 40, b"zz", b"abab", b"\n", b"b\n", b"a" * 19_999, b"ab" * 2]
    pats = [base[i % len(base)] for i in range(5000)]
    with gpu.DeviceIndex(text.size, 0) as idx:
        idx.build(text)
        idx.set_rows(starts)
        for k in (1, 3, 4, 13, 64):
            monkeypatch.setenv("SA_HIP_HOST_ROWS", "1")
            host_rows, host_rg = idx.query_rows_batch(pats[:len(base)], k)
            monkeypatch.delenv("SA_HIP_HOST_ROWS")
            dev_rows, dev_rg = idx.query_rows_batch(pats, k)
            monkeypatch.setenv("SA_HIP_ROWS_WAVES", "0")
            nw_rows, nw_rg = idx.query_rows_batch(pats, k)
            monkeypatch.delenv("SA_HIP_ROWS_WAVES")
            assert np.array_equal(dev_rg, nw_rg) and np.array_equal(dev_rg[:len(base)], host_rg)
            for i, (a, b) in enumerate(zip(dev_rows, nw_rows)):
                assert np.array_equal(a, b), (k, pats[i], a, b)
                assert np.array_equal(a, host_rows[i % len(base)]), (k, pats[i], a, host_rows[i % len(base)])
        rows, _ = idx.query_rows_batch(pats[:len(base)] * 400, 13)
        assert sorted(rows[0].tolist()) == [0, 1, 2, 3, 4] and sorted(rows[3].tolist()) == [0, 1, 2, 3, 4] and rows[8].size == 0


def test_api_edge_cases(gpu):
    lib = gpu.lib()
    import ctypes as C
    h = C.c_void_p()
    assert lib.sa_hip_index_create(C.byref(h), 1 << 33, 0) == -1          # beyond 2^32 - 2
    assert lib.sa_hip_index_create(C.byref(h), 16, 99) == -3              # no such device
    assert lib.sa_hip_index_create(None, 16, 0) == -1
    with gpu.DeviceIndex(0, 0) as idx:                                    # zero-capacity index
        idx.build(np.zeros(0, np.uint8))
        assert idx.n == 0 and idx.verify() == 0
        with pytest.raises(gpu.SaHipError):
            idx.build(np.frombuffer(b"too long", np.uint8))               # exceeds the capacity
    with gpu.DeviceIndex(100, 0) as idx:
        with pytest.raises(gpu.SaHipError):
            idx.query_batch([b"x"])                                       # no index yet
        with pytest.raises(gpu.SaHipError):
            idx.sa_u32()
        idx.build(np.frombuffer(b"abracadabra", np.uint8))
        assert idx.sa_u32().tolist() == [10, 7, 0, 3, 5, 8, 1, 4, 6, 9, 2]
        with pytest.raises(gpu.SaHipError):
            idx.sa_range(5, 100)
        assert idx.sa_range(0, 0).size == 0
        assert b"no index" not in lib.sa_hip_last_error() or True


def test_directory_from_flags_pass_matches_searched_directory(gpu, oracle, monkeypatch):
    """The bucket directory written by the first flags pass of a build (runs of buckets between
    consecutive keys; long runs queued for dir_fill_kernel) against the directory built by binary
    search (SA_HIP_FUSE_DIR=0) and against the oracle, on alphabets that leave large holes in the key space."""
    rng = np.random.default_rng(77)
    texts = {
        "two_far_bytes": rng.choice(np.array([1, 254], dtype=np.uint8), 300_000),
        "sparse_5": rng.choice(np.array([3, 9, 10, 200, 255], dtype=np.uint8), 200_000),
        "uniform27": rng.integers(97, 124, 1_000_000).astype(np.uint8),
        "one_symbol_run_then_noise": np.concatenate([np.full(50_000, 120, np.uint8), rng.integers(97, 100, 50_000).astype(np.uint8)]),
        "tiny": np.frombuffer(b"abracadabra", dtype=np.uint8).copy(),
    }
    for name, t in texts.items():
        pats = cases.query_patterns(t, 400, rng)
        got = {}
        for fuse in ("1", "0"):
            monkeypatch.setenv("SA_HIP_FUSE_DIR", fuse)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, 0)
                got[fuse] = idx.query_batch(pats)
        assert np.array_equal(got["1"], got["0"]), name
        if t.size <= 300_000:
            sa = oracle.sais(t).astype(np.uint32)
            assert np.array_equal(got["1"], oracle.query_batch(t, sa, 0xFFFFFFFF, pats)), name


def test_query_hits_equals_batch_plus_range(gpu, oracle):
    """sa_hip_index_query_hits (one query + its first hits through a pinned block) against the batched query and
    sa_hip_index_get_sa_range, hits and misses, the all-smaller sentinel, an empty pattern, hit counts beyond the cap."""
    rng = np.random.default_rng(5)
    t = rng.integers(97, 101, 200_000).astype(np.uint8)   # 4 symbols: short patterns have many thousand hits
    pats = cases.query_patterns(t, 200, rng, maxlen=12) + [b"a", b"ab", b"zzz", b"", bytes([96])]
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.build(t)
        exp = idx.query_batch(pats)
        for p, (f, s) in zip(pats, exp):
            for cap in (1, 100, 4096, 10_000):
                (f2, s2), hits = idx.query_hits(p, cap)
                assert (f2, s2) == (int(f), int(s)), p
                count = 0 if (f == 0xFFFFFFFF or ((int(s) - int(f) + 1) & 0xFFFFFFFF) == 0) else int(s) - int(f) + 1
                take = min(count, cap, 4096)
                assert hits.size == take, (p, cap)
                if take:
                    assert np.array_equal(hits, idx.sa_range(int(f), take))


def test_narrow_key_array_matches_wide_key_array(gpu, oracle, monkeypatch):
    """After a narrow-record sort the index keeps u32 narrow keys + 257 bucket bounds as its query key array
    (SA_HIP_NARROW_K, default on) instead of rebuilt u64 keys: same suffix array, same directory, same ranges --
    against the u64 form and against the oracle (engine.c:869-918).  Patterns of 1..40 bytes: one- and
    two-character patterns span several top digits, long ones go on into the text comparison; a skewed alphabet
    leaves most top-digit buckets empty; forced key lengths give narrow keys of fewer than 32 bits."""
    from suffixarray_amd import synth
    rng = np.random.default_rng(5)
    skew = rng.choice(np.array([97, 98, 99, 100, 122], dtype=np.uint8), 5_000_000, p=[0.9, 0.04, 0.03, 0.02, 0.01])
    two = rng.choice(np.array([97, 122], dtype=np.uint8), 4_500_000)
    runs = [(synth.d1_uniform27(4_500_001), 0, 0), (synth.d1_uniform27(5_000_000), 7, 0), (synth.d1_uniform27(5_000_000), 0, 32),
            (skew, 13, 0), (skew, 5, 0), (two, 20, 0), (two, 9, 6), (synth.d2_words(6_000_000), 8, 0),
            (rng.integers(0, 256, 4_500_000).astype(np.uint8), 0, 4)]
    for t, k0, L in runs:
        if k0:
            monkeypatch.setenv("SA_HIP_INITIAL_CHARS", str(k0))
        else:
            monkeypatch.delenv("SA_HIP_INITIAL_CHARS", raising=False)
        pats = cases.query_patterns(t, 3000, rng)
        pats += [bytes([c]) for c in np.unique(t)[:8]] + [bytes(t[p:p + m]) for p in (0, 17, t.size - 9) for m in (1, 2, 3, 8, 9)]
        got, sas = {}, {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SA_HIP_NARROW_K", mode)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                st = idx.build_stats()
                assert st["narrow_k"] == (1 if mode == "1" else 0), st
                assert st["pass_launches"][3] == 1, st
                assert idx.verify() == 0, st
                sas[mode] = idx.sa_u32().copy()
                got[mode] = idx.query_batch(pats)
        assert np.array_equal(sas["1"], sas["0"]), (t.size, k0, L)
        assert np.array_equal(got["1"], got["0"]), (t.size, k0, L)
        exp = oracle.query_batch(t, sas["1"], L if L else 0xFFFFFFFF, pats)
        assert np.array_equal(got["1"], exp), (t.size, k0, L)


def test_sector_search_matches_binary_search(gpu, oracle, monkeypatch):
    """Inside a directory bucket the key search is an interpolated scan of 32-byte windows (sa_query.hpp: sector_bound;
    default, SA_HIP_SECTOR_SEARCH=2) or whole 64-byte sectors (=1) instead of a binary search (=0): identical ranges, equal to the oracle's
    (engine.c:869-918), for narrow (u32) and wide (u64) key arrays, adopted indexes (keys gathered from the text), a
    skewed alphabet (buckets of very different sizes, long runs of equal keys: the estimate is far off and the scan falls
    back to bisection), word text (equal keys in the thousands), texts shorter than a sector, truncated indexes."""
    from suffixarray_amd import synth
    rng = np.random.default_rng(99)
    skew = rng.choice(np.array([97, 98, 99, 100, 122], dtype=np.uint8), 5_000_000, p=[0.9, 0.04, 0.03, 0.02, 0.01])
    st = cases.small_texts()
    runs = [("d1", synth.d1_uniform27(4_600_003), 0, {}), ("d1_wide", synth.d1_uniform27(4_600_003), 0, {"SA_HIP_NARROW_K": "0"}),
            ("d1_L12", synth.d1_uniform27(4_500_000), 12, {}), ("skew", skew, 0, {}), ("skew_k5", skew, 0, {"SA_HIP_INITIAL_CHARS": "5"}),
            ("words", synth.d2_words(5_000_000), 0, {}), ("words_narrow", synth.d2_words(5_000_000), 0, {"SA_HIP_PILOT": "0"}),
            ("banana", st["banana"], 0, {}), ("r27_63", st["r27_63"], 0, {}), ("all_a_5000", st["all_a_5000"], 0, {}),
            ("d2_300k", st["d2_300k"], 0, {}), ("d2_300k_L7", st["d2_300k"], 7, {})]
    for name, t, L, env in runs:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pats = cases.query_patterns(t, 4000, rng)
        pats += [bytes([c]) for c in np.unique(t)[:8]] + [bytes(t[p:p + m]) for p in (0, max(t.size - 9, 0)) for m in (1, 2, 3, 8, 9, 16)]
        got = {}
        sa = None
        for mode in ("2", "1", "0"):
            monkeypatch.setenv("SA_HIP_SECTOR_SEARCH", mode)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                sa = idx.sa_u32().copy()
                got[mode] = idx.query_batch(pats)
                idx.load(t, sa, L)           # adopted: u64 keys gathered from the text, directory by binary search
                got[mode + "_adopted"] = idx.query_batch(pats)
        for k in env:
            monkeypatch.delenv(k, raising=False)
        exp = oracle.query_batch(t, sa, L if L else 0xFFFFFFFF, pats)
        for k, v in got.items():
            assert np.array_equal(v, exp), (name, k)
    monkeypatch.delenv("SA_HIP_SECTOR_SEARCH", raising=False)


def test_fixed_length_batch_equals_offsets_batch(gpu, oracle):
    """sa_hip_query_batch_device_fixed (Q patterns of one length, no offsets array -- what bench.py calls) against
    sa_hip_query_batch_device and the oracle, for lengths around the word size of the pattern loads and beyond the key."""
    import torch
    from suffixarray_amd import synth
    rng = np.random.default_rng(8)
    for t, L in ((synth.d1_uniform27(4_600_000), 0), (synth.d2_words(3_000_000), 0), (synth.d2_words(3_000_000), 20)):
        with gpu.DeviceIndex(t.size, 0) as idx:
            idx.build(t, L)
            sa = idx.sa_u32()
            for m in (1, 2, 7, 8, 9, 16, 31, 32, 33, 40):
                q = 3000
                pos = rng.integers(0, t.size - m, q)
                pats = np.stack([t[p:p + m] for p in pos])
                pats[1::2] = rng.integers(97, 123, (q // 2, m), dtype=np.uint8)
                flat = np.ascontiguousarray(pats.reshape(-1))
                off = (np.arange(q + 1, dtype=np.uint64) * np.uint64(m))
                exp = oracle.query_batch(t, sa, L if L else 0xFFFFFFFF, (flat, off))
                d_pat = torch.from_numpy(np.concatenate([flat, np.zeros(64, np.uint8)])).to("cuda:0")
                d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
                out_a = torch.zeros(2 * q, dtype=torch.int32, device="cuda:0")
                out_b = torch.zeros(2 * q, dtype=torch.int32, device="cuda:0")
                torch.cuda.synchronize()
                idx.query_batch_device_fixed(d_pat.data_ptr(), m, q, out_a.data_ptr())
                idx.query_batch_device(d_pat.data_ptr(), d_off.data_ptr(), q, out_b.data_ptr())
                idx.sync()
                a = out_a.cpu().numpy().view(np.uint32).reshape(-1, 2)
                b = out_b.cpu().numpy().view(np.uint32).reshape(-1, 2)
                assert np.array_equal(a, b), (m, L)
                assert np.array_equal(a[:, 0], exp["first"]) and np.array_equal(a[:, 1], exp["second"]), (m, L)


def test_deep_keys_give_the_same_ranges(gpu, oracle, tmp_path, monkeypatch):
    """Second-level keys (round 4, sa_hip_index_prepare_deep_keys: the characters that follow the key, for the slots that share
    their key with a neighbour): a pattern longer than the key finds its bounds inside a key group by a search over 8-byte keys
    instead of text comparisons.  Name-like and word-like texts, full and truncated builds, an adopted index: the ranges with
    and without them are identical and equal the oracle's -- whole names (all hit, many with thousands of hits), every prefix
    length around the key and the second key (10..24 characters), names with a changed last character (misses inside a key
    group), patterns longer than both keys, patterns with a byte that does not occur, patterns past the end of the text."""
    from suffixarray_amd import synth
    path = str(tmp_path / "companies.csv")
    gpu.synth_csv(path, 60_000, 3)
    names = np.array(gpu.csv_extract_column(path, "company_name", copy=False)[1])
    words = synth.d2_words(1_200_000)
    rng = np.random.default_rng(11)

    def patterns(t):
        ends = np.flatnonzero(t == 10)
        pats = []
        for j in rng.integers(1, ends.size, 1500):
            a, b = int(ends[j - 1]) + 1, int(ends[j])
            row = bytes(t[a:b])
            if not row:
                continue
            pats.append(row)                                   # a whole row (name / line): hits
            pats.append(row[:int(rng.integers(1, len(row) + 1))])   # a prefix of any length
            if len(row) > 12:
                pats.append(row[:-1] + bytes([(row[-1] + 1) & 0x7F or 65]))   # differs at its last character
                pats.append(row[3:])                           # a suffix of the row: still a substring of the text
            pats.append(row + b"\n" + bytes(t[b + 1:b + 1 + int(rng.integers(0, 20))]))   # across the row boundary: longer than both keys
        for m in range(9, 26):                                 # windows of every length around the two key lengths
            for p in rng.integers(0, t.size - 40, 40):
                pats.append(bytes(t[p:p + m]))
        pats += [b"zzzzzzzzzzzzzzzzzzzzzzzz", b"INTERNATIONAL \x01X", b"\xff" * 13, bytes(t[-5:]), bytes(t[-30:]) + b"tail", b"A"]
        return pats

    for name, t in (("names", names), ("words", words)):
        pats = patterns(t)
        for L in (32, 0, 15):
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                st = idx.build_stats()
                sa = idx.sa_u32().copy()
                exp = oracle.query_batch(t, sa, L if L else 0xFFFFFFFF, pats)
                plain = idx.query_batch(pats)
                has = idx.prepare_deep_keys()
                deep = idx.query_batch(pats)
                assert np.array_equal(plain, exp), (name, L)
                assert np.array_equal(deep, exp), (name, L, has)
                assert has == (st["narrow_k"] == 0), (name, L, st["narrow_k"])   # wide keys: built; narrow keys: no use for them
                # an adopted index (keys gathered from the text) answers the same way with them
                if L == 32:
                    with gpu.DeviceIndex(t.size, 0) as idx2:
                        idx2.load(t, sa, L)
                        assert idx2.prepare_deep_keys()
                        assert np.array_equal(idx2.query_batch(pats), exp), (name, "adopted")
    # a batch of at least 32768 patterns builds them on its way; the rebuilt index drops them again.  From 2^20 patterns on a
    # batch over a wide-key index is also answered in the order of its patterns' first characters (two counting passes, then
    # thread i answers query perm[i]): the threshold lowered for this test -- the same ranges, in the batch's own order
    big = patterns(names)
    big = (big * (32768 // len(big) + 1))[:40000]
    with gpu.DeviceIndex(names.size, 0) as idx:
        idx.build(names, 32)
        sa = idx.sa_u32().copy()
        exp = oracle.query_batch(names, sa, 32, big)
        got = idx.query_batch(big)
        assert np.array_equal(got, exp)
        monkeypatch.setenv("SA_HIP_QCLUSTER_MIN", "4096")
        assert np.array_equal(idx.query_batch(big), exp)
        assert np.array_equal(idx.query_batch(big[:5000]), exp[:5000])   # a last tile that is not full
        monkeypatch.delenv("SA_HIP_QCLUSTER_MIN")
        idx.build(names, 32)
        assert np.array_equal(idx.query_batch(big[:500]), got[:500])
