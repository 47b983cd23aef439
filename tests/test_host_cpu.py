"""CPU: the C ABI loads and exports every declared symbol (no compute without a GPU), the
pipeline model (host logic of the device build) matches the oracle, host helpers behave."""
import os
import re

import numpy as np
import pytest

import cases
from conftest import GOLDEN
from pipeline_model import build_sa_model, three_pass_model, choose_split_level
from test_oracle import check_truncated_order


def test_library_exports_every_declared_symbol(capi):
    lib = capi.lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "sa_hip.h")).read()
    declared = set(re.findall(r"\b(sa_hip_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(capi.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert b"gfx950" in lib.sa_hip_version()


def test_struct_layouts_match_reference_abi(capi):
    import ctypes as C
    assert C.sizeof(capi.SuffixArrayStruct) == 40   # engine.h:123-130
    assert C.sizeof(capi.PairU32) == 8              # engine.h:219-222


def test_stats_structs_match_the_header(capi, tmp_path):
    """The ctypes mirrors of the instrumentation structs against the C compiler's view of include/sa_hip.h."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sa_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sa_hip_build_stats), offsetof(sa_hip_build_stats, radix_ms),'
                   ' offsetof(sa_hip_build_stats, pass_ms), offsetof(sa_hip_build_stats, pass_launches), sizeof(sa_hip_query_stats),'
                   ' sizeof(sa_hip_csv_column), sizeof(sa_hip_big_stats), offsetof(sa_hip_big_stats, tied_after_sort),'
                   ' offsetof(sa_hip_big_stats, total_ms)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    B = capi.BuildStats
    G = capi.BigStats
    assert got == [C.sizeof(B), B.radix_ms.offset, B.pass_ms.offset, B.pass_launches.offset, C.sizeof(capi.QueryStats),
                   C.sizeof(capi.CsvColumn), C.sizeof(G), G.tied_after_sort.offset, G.total_ms.offset], got


def test_no_gpu_means_loud_failure(capi):
    """Without a device every compute entry point must fail, never fall back."""
    if capi.lib().sa_hip_device_count() >= 1:
        pytest.skip("a HIP device is present")
    with pytest.raises(capi.SaHipError):
        capi.DeviceIndex(1024, 0)
    with pytest.raises(capi.SaHipError):
        capi.libsais(b"banana")
    import numpy as np
    buf = np.zeros(64, np.uint8)
    with pytest.raises(capi.SaHipError):   # the 64-bit-index build and its sufcheck (round 4) refuse as loudly
        capi.libsais64_device(buf.ctypes.data, buf.ctypes.data, 4)
    with pytest.raises(capi.SaHipError):
        capi.sufcheck64_device(buf.ctypes.data, buf.ctypes.data, 4)


def test_config1_readme_on_the_opt_in_host_path(capi, oracle, monkeypatch, tmp_path):
    """BASELINE config 1 ("3-doc README example via SuffixArray(documents=...), CPU path, no GPU"): with no HIP device AND
    SA_HIP_ALLOW_HOST=1 the handle API is served by the library's own small host implementation (csrc/host_index.hpp;
    never the oracle, which only checks it here).  The README documents give the reference-generated golden ranges; suffix
    arrays equal the oracle's on small and adversarial texts, full and truncated; record retrieval, the CSV mode and
    save / load run on it; without the variable nothing changes (test above)."""
    if capi.lib().sa_hip_device_count() >= 1:
        pytest.skip("a HIP device is present: the host path is never taken")
    monkeypatch.setenv("SA_HIP_ALLOW_HOST", "1")
    from suffixarray_amd import SuffixArray
    g = np.load(os.path.join(GOLDEN, "golden_readme.npz"), allow_pickle=False)
    L = int(g["max_suffix_length"][0])
    docs = ["The quick brown fox jumps over the lazy dog", "I am going to the store to buy some milk",
            "Uhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhh"]   # README.md:16-20
    assert "\n".join(docs).lower().encode() == bytes(g["text"])
    s = SuffixArray(documents=docs, max_suffix_length=L)
    pats = [bytes(p) for p in g["patterns"]]
    got = s.query_ranges(pats)
    assert np.array_equal(got["first"], g["ranges"]["first"]) and np.array_equal(got["second"], g["ranges"]["second"])
    assert s.query_records("the quick brown fox") == [docs[0]]
    assert sorted(s.query_records("THE")) == sorted(docs[:2]) and s.query_records("zzz") == [] and s.query_records("milk", k=1) == [docs[1]]
    assert s.query_records_batch(["uhh", "", "the"], k=1000)[0] == [docs[2]]
    idx = s._index
    assert idx.verify() == 0 and np.array_equal(idx.sa_u32(), oracle.truncated_sa(g["text"], L))
    s.save(str(tmp_path / "idx"))
    s2 = SuffixArray.load(str(tmp_path / "idx"))
    assert s2.query_records("milk") == [docs[1]]
    s.close(); s2.close()
    # the handle API against the oracle: full and truncated order, query conventions, the libsais-compatible calls
    rng = np.random.default_rng(3)
    texts = cases.small_texts()
    for name in ("banana", "mississippi", "len1", "aa", "all_a_5000", "ab_3000", "fib", "highbit", "with_nul", "r27_4097", "r2_30000", "repeat_block", "d2_300k"):
        t = texts[name]
        sa = oracle.sais(t).astype(np.uint32)
        with capi.DeviceIndex(t.size, 0) as idx:
            idx.build(t)
            assert np.array_equal(idx.sa_u32(), sa), name
            assert np.array_equal(idx.sa_i64(), sa.astype(np.int64)) and idx.verify() == 0
            pats = cases.query_patterns(t, 200, rng)
            assert np.array_equal(idx.query_batch(pats), oracle.query_batch(t, sa, 0xFFFFFFFF, pats)), name
            for Lt in (1, 2, 3, 5, 8, 13, 32):
                idx.build(t, Lt)
                assert np.array_equal(idx.sa_u32(), oracle.truncated_sa(t, Lt)), (name, Lt)
                assert idx.verify() == 0
                assert np.array_equal(idx.query_batch(pats[:50]), oracle.query_batch(t, idx.sa_u32(), Lt, pats[:50])), (name, Lt)
            with pytest.raises(capi.SaHipError):
                idx.build_device(0, t.size)          # no device buffers on the host path
        if t.size < 70000:
            assert np.array_equal(capi.libsais(t), sa.astype(np.int32)) and np.array_equal(capi.libsais64(t), sa.astype(np.int64))
    with pytest.raises(capi.SaHipError):
        capi.DeviceIndex((1 << 24) + 1, 0)           # the host path is small on purpose
    cases.check_partitioned(SuffixArray, tmp_path)
    # CSV mode end to end
    p = tmp_path / "c.csv"
    p.write_text('id,company_name,country\n1,Netflix,US\n2,"Acme, Inc.",US\n3,netflix studios,US\n')
    c = SuffixArray(csv_file=str(p), search_column="company_name", max_suffix_length=32)
    assert [r["id"] for r in c.query_records("netflix")] in (["1", "3"], ["3", "1"]) and c.query_records("acme, inc")[0]["country"] == "US"
    c.close()


@pytest.mark.parametrize("flags", ["-fsanitize=address,undefined -fno-sanitize-recover=all", "-fsanitize=thread"])
def test_host_code_under_sanitizers(tmp_path, flags):
    """The host-only C++ of the library -- the multi-threaded RFC-4180 column extractor, row copying, hits -> rows, the
    opt-in host index -- compiled by the host compiler with AddressSanitizer + UBSan, and with ThreadSanitizer, and run
    (tools/host_sanitize.cpp): sanitizers belong on a CPU build, never on the GPU pool."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no host compiler / HIP headers")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", *flags.split(), "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(root, "suffixarray_amd", "csrc"), os.path.join(root, "tools", "host_sanitize.cpp"), "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and ("cannot find" in r.stderr or "unrecognized" in r.stderr):
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_reference_file_layout_roundtrip(capi, tmp_path):
    """sa_hip_write_suffix_array / sa_hip_read_suffix_array use the reference's own layout (engine.c:1123-1128:
    {u64 start, u64 end, u32 max_suffix_length, u32 n, u32 SA[n]}; the bit-buffer file {u32 capacity, bytes}, engine.c:1098-1101)."""
    import ctypes as C
    sa = np.array([5, 3, 1, 0, 4, 2], dtype=np.uint32)
    st = capi.SuffixArrayStruct()
    st.suffix_array = sa.ctypes.data
    st.global_byte_start_idx, st.global_byte_end_idx, st.max_suffix_length, st.n = 7, 1234567890123, 32, sa.size
    f, q = str(tmp_path / "sa.bin"), str(tmp_path / "q.bin")
    capi.check(capi.lib().sa_hip_write_suffix_array(C.byref(st), f.encode(), q.encode()))
    raw = open(f, "rb").read()
    assert len(raw) == 24 + 4 * sa.size
    assert np.frombuffer(raw[:16], "<u8").tolist() == [7, 1234567890123] and np.frombuffer(raw[16:24], "<u4").tolist() == [32, 6]
    assert np.array_equal(np.frombuffer(raw[24:], "<u4"), sa)
    assert open(q, "rb").read() == b"\0\0\0\0"
    back = capi.SuffixArrayStruct()
    capi.check(capi.lib().sa_hip_read_suffix_array(C.byref(back), f.encode()))
    assert (back.n, back.max_suffix_length, back.global_byte_start_idx, back.global_byte_end_idx) == (6, 32, 7, 1234567890123)
    assert np.ctypeslib.as_array(C.cast(back.suffix_array, C.POINTER(C.c_uint32)), (6,)).tolist() == sa.tolist()
    capi.lib().sa_hip_free_suffix_array(C.byref(back))
    open(f, "wb").write(raw[:-3])
    assert capi.lib().sa_hip_read_suffix_array(C.byref(back), f.encode()) == -1 and not back.suffix_array
    bad = bytearray(raw); bad[24:28] = (99).to_bytes(4, "little")
    open(f, "wb").write(bytes(bad))
    assert capi.lib().sa_hip_read_suffix_array(C.byref(back), f.encode()) == -1
    assert capi.lib().sa_hip_read_suffix_array(C.byref(back), str(tmp_path / "missing").encode()) == -1


def test_synth_d1_matches_definition(capi):
    a = capi.synth_uniform27(1000)
    s = 88172645463325252
    m = (1 << 64) - 1
    exp = []
    for _ in range(1000):
        s ^= (s << 13) & m
        s ^= s >> 7
        s ^= (s << 17) & m
        v = (s >> 33) % 27
        exp.append(10 if v == 26 else 97 + v)
    assert a.tolist() == exp


def test_pipeline_model_full(oracle):
    for name, t in cases.small_texts().items():
        if t.size > 70000 or t.size == 0:
            continue
        exp = oracle.sais(t).astype(np.uint32)
        for cr in (0, 2):
            got = build_sa_model(t, chunk_rounds_before_doubling=cr)
            assert np.array_equal(got, exp), (name, cr)


def test_pipeline_model_truncated(oracle):
    for name, t in cases.small_texts().items():
        if t.size > 20000 or t.size == 0:
            continue
        for L in (1, 2, 5, 32, 64):
            got = build_sa_model(t, max_suffix_length=L)
            assert np.array_equal(got, oracle.truncated_sa(t, L)), (name, L)
            check_truncated_order(t, got, L)


def test_three_pass_plan_model():
    """The host logic of the three-pass plan (csrc/radix_split.hpp) and the identities its local pass rests on, in numpy: the
    level rule (smallest level that fits, then the large form, else declined), records grouped in ANY order and ordered by
    (key, suffix) per sub-bucket = the stable sort; the directory slice of a sub-bucket = its bin-start table subsampled = the
    directory by binary search; the slots staged from inside the sub-buckets = the slots whose key equals a neighbour's."""
    rng = np.random.default_rng(21)
    assert choose_split_level([10 ** 6, 9000, 8192, 4000], 32) == (2, False)
    assert choose_split_level([10 ** 6, 20000, 16000, 9000], 32) == (2, True)
    assert choose_split_level([10 ** 6, 20000, 16000, 9000], 13) == (None, False)      # (no 12 key bits below the sub-bucket)
    assert choose_split_level([10 ** 6, 10 ** 5, 10 ** 5, 10 ** 5], 32) == (None, False)
    for n, sigma, k0, cap, dbits in ((60000, 27, 6, 64, 19), (60000, 27, 5, 300, 17), (50000, 4, 12, 40, 19), (40000, 27, 6, 16, 19)):
        b = 1
        while (1 << b) < sigma + 1:
            b += 1
        codes = rng.integers(1, sigma + 1, n + k0).astype(np.uint64)
        codes[n:] = 0
        key = np.zeros(n, np.uint64)
        for j in range(k0):
            key = (key << np.uint64(b)) | codes[np.arange(n) + j]
        lo_bits = b * k0 - 8
        m = three_pass_model(key, lo_bits, dbits, cap=cap, cap_big=2 * cap, seed=n)
        assert m is not None and m["levels"][m["rb"]] <= (2 * cap if m["big"] else cap), (n, sigma, k0)
        assert m["rb"] == 1 or m["levels"][m["rb"] - 1] > cap
        order = np.argsort(key, kind="stable")
        assert np.array_equal(m["sa"], order) and np.array_equal(m["keys"], key[order])
        top = (key[order] >> np.uint64(b * k0 - dbits)).astype(np.int64)
        assert np.array_equal(m["dir"][:-1], np.searchsorted(top, np.arange(1 << dbits), side="left")) and m["dir"][-1] == n
        ks = key[order]
        eq_prev = np.concatenate([[False], ks[1:] == ks[:-1]])
        eq_next = np.concatenate([ks[1:] == ks[:-1], [False]])
        exp = sorted((int(p), bool(not eq_prev[p])) for p in np.flatnonzero(eq_prev | eq_next))
        assert m["staged"] == exp, (n, sigma, k0)
    # skew: one symbol nine times in ten -- no level fits, the plan is declined
    key = np.zeros(40000, np.uint64)
    codes = rng.choice(np.array([1, 2, 3], np.uint64), 40000 + 16, p=[0.9, 0.05, 0.05])
    for j in range(16):
        key = (key << np.uint64(2)) | codes[np.arange(40000) + j]
    assert three_pass_model(key, 24, 14, cap=64, cap_big=128) is None


def test_csv_native_matches_python_reference(tmp_path, capi):
    from csv_ingest import extract_column, extract_column_py
    p = tmp_path / "t.csv"
    p.write_bytes(b'id,name,country\n1,Netflix,US\n2,"Acme, Inc.",US\n\n3,"Multi\nLine ""Q""",DE\r\n4,netflix studios,US\n5,,FR\n6')
    big = tmp_path / "big.csv"
    capi.synth_csv(str(big), 20000, 3)
    for path, col in ((p, "name"), (p, "country"), (big, "company_name"), (big, "id")):
        a, b = extract_column(str(path), col), extract_column_py(str(path), col)
        assert a.columns == b.columns
        assert a.text == b.text
        assert np.array_equal(a.text_row_starts, b.text_row_starts)
        assert np.array_equal(a.row_file_offsets, b.row_file_offsets)
    data = big.read_bytes()
    assert data.startswith(b"id,company_name,country\n") and b'", Inc."' not in data and b', Inc."' in data


def test_csv_native_matches_python_reference_1e6_rows(tmp_path, capi, monkeypatch):
    """The native extractor pinned at a size where its range splitting is really at work (1.2e6 rows of the config-5 generator,
    ~34 MB: several worker parts, quoted ', Inc.' fields across part boundaries) against the byte-serial Python state machine
    (tests/csv_ingest.py: extract_column_py).  tests/test_gpu_full_size.py::test_config5_full_size_csv_mode takes its expected
    column from the native extractor; this is what pins that extractor beyond golden_csv.npz's 10 000 rows."""
    from csv_ingest import extract_column, extract_column_py
    big = tmp_path / "big.csv"
    capi.synth_csv(str(big), 1_200_000, 7)
    ref = extract_column_py(str(big), "company_name")
    assert len(ref.text_row_starts) == 1_200_000
    for threads in ("1", "5", "16"):
        monkeypatch.setenv("SA_HIP_CSV_THREADS", threads)
        got = extract_column(str(big), "company_name")
        assert got.columns == ref.columns
        assert got.text == ref.text, threads
        assert np.array_equal(got.text_row_starts, ref.text_row_starts), threads
        assert np.array_equal(got.row_file_offsets, ref.row_file_offsets), threads


def test_csv_native_randomized_against_python_reference(tmp_path, capi, monkeypatch):
    """Random small CSV files built from the pieces that steer the two native paths (rows without a quote character go
    through memchr sweeps, the others through the state machine): bare CR, CRLF, blank lines, rows with fewer fields than
    the wanted column, quoted fields with commas / newlines / doubled quotes, a last row without a terminator; 1 and 3
    worker threads (tiny files give one part; the range splitting is covered by the 20 000-row file above)."""
    from csv_ingest import extract_column, extract_column_py
    rng = np.random.default_rng(17)
    atoms = [b"abc", b"X", b"", b"Hello World", b'"q,1"', b'"two\nlines"', b'"say ""hi"""', b'""', b"a b", b"Z9"]
    eols = [b"\n", b"\n", b"\n", b"\r\n", b"\r", b"\n\n", b"\r\n\r\n"]
    p = tmp_path / "r.csv"
    for trial in range(150):
        rows = [b"c0,c1,c2"]
        for _ in range(int(rng.integers(0, 12))):
            nf = int(rng.integers(1, 5))
            rows.append(b",".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(nf)))
        data = b""
        for r in rows:
            data += r + eols[int(rng.integers(0, len(eols)))]
        if trial % 3 == 0:
            data = data.rstrip(b"\r\n") if len(rows) > 1 else data   # last row without a terminator
        p.write_bytes(data)
        for threads in ("1", "3"):
            monkeypatch.setenv("SA_HIP_CSV_THREADS", threads)
            for col in ("c0", "c1", "c2"):
                a, b = extract_column(str(p), col), extract_column_py(str(p), col)
                assert a.text == b.text, (trial, col, data)
                assert np.array_equal(a.text_row_starts, b.text_row_starts), (trial, col, data)
                assert np.array_equal(a.row_file_offsets, b.row_file_offsets), (trial, col, data)


def test_csv_ingest(tmp_path):
    from csv_ingest import extract_column
    p = tmp_path / "c.csv"
    p.write_text('id,name,country\n1,Netflix,US\n2,"Acme, Inc.",US\n3,"Multi\nLine ""Q""",DE\n4,netflix studios,US\n')
    col = extract_column(str(p), "name")
    assert col.columns == ["id", "name", "country"]
    assert col.text == b'netflix\nacme, inc.\nmulti line "q"\nnetflix studios\n'
    assert col.text_row_starts.tolist() == [0, 8, 19, 34]
    data = p.read_bytes()
    assert data[col.row_file_offsets[1]:col.row_file_offsets[2]] == b'2,"Acme, Inc.",US\n'
    with pytest.raises(ValueError):
        extract_column(str(p), "nope")


def test_cython_binding_builds_and_fails_loudly_without_gpu(capi):
    from suffixarray_amd.suffix_array import SuffixArray
    if capi.lib().sa_hip_device_count() >= 1:
        pytest.skip("a HIP device is present")
    with pytest.raises(RuntimeError):
        SuffixArray(documents=["a", "b"], max_suffix_length=8)
    with pytest.raises(ValueError):
        SuffixArray(documents=["a"], csv_file="x.csv")


def test_lifecycle_mirrors_of_the_seam(capi):
    """sa_hip_init_suffix_array_byte_idxs / sa_hip_free_suffix_array (engine.c:326-349): host-side malloc / free of the
    caller's uint32 array, no device involved."""
    import ctypes as C
    st = capi.SuffixArrayStruct()
    assert capi.lib().sa_hip_init_suffix_array_byte_idxs(C.byref(st), 32, 5, 105, 100) == 0
    assert st.suffix_array and st.n == 100 and st.max_suffix_length == 32
    assert st.global_byte_start_idx == 5 and st.global_byte_end_idx == 105 and not st.is_quoted_bitflag
    C.memset(st.suffix_array, 0xAB, 400)   # writable for n entries
    capi.lib().sa_hip_free_suffix_array(C.byref(st))
    assert not st.suffix_array
    capi.lib().sa_hip_free_suffix_array(C.byref(st))   # idempotent
    assert capi.lib().sa_hip_init_suffix_array_byte_idxs(None, 32, 0, 0, 1) == -1


def test_csv_extractor_against_reference_row_sets(capi, tmp_path):
    """The native column extractor pinned to the reference's CSV builder: on the files of
    tests/golden/golden_csv.npz the rows whose extracted field contains a pattern, and the number of occurrences,
    equal what the compiled reference returned (get_substring_positions_file / get_matching_records_file over
    construct_truncated_suffix_array_from_csv_partitioned_mmap) for every pattern without a byte below ','."""
    import os
    from conftest import GOLDEN
    from csv_ingest import extract_column
    g = np.load(os.path.join(GOLDEN, "golden_csv.npz"), allow_pickle=False)
    for name in g["names"]:
        data = bytes(g[f"csv__{name}"])
        column = str(g[f"column__{name}"][0])
        path = tmp_path / f"{name}.csv"
        path.write_bytes(data)
        col = extract_column(str(path), column)
        fields = col.text.split(b"\n")[:-1]
        # the reference's text = ours + the header row's field (it indexes the header like a record)
        assert len(col.text) == int(g[f"n__{name}"][0]) - (len(column) + 1)
        ids = [int(data[int(a):int(b)].split(b",", 1)[0]) for a, b in zip(col.row_file_offsets[:-1], col.row_file_offsets[1:])]
        pats, off, rid = g[f"patterns__{name}"], g[f"row_ids_offsets__{name}"], g[f"row_ids__{name}"]
        n_checked = 0
        for i, p in enumerate(pats):
            if not bool(g[f"letters_only__{name}"][i]):
                continue
            pb = str(p).encode()
            rows = sorted(j for j, f in zip(ids, fields) if pb in f)
            occ = sum(sum(1 for o in range(len(f)) if f.startswith(pb, o)) for f in fields)
            assert rows == rid[off[i]:off[i + 1]].tolist(), (name, p)
            assert occ == int(g[f"hit_counts__{name}"][i]) == int(g[f"record_counts__{name}"][i]), (name, p)
            n_checked += 1
        assert n_checked >= 5


def test_record_seam_argument_validation(capi, tmp_path):
    """The record-retrieval entry points return codes instead of crashing or exiting (the reference printf()s and
    exit(1)s: engine.c:1334-1338, 475-483): NULL arguments, a missing file, an unknown column -- all without a device."""
    import ctypes as C
    L = capi.lib()
    h = C.c_void_p()
    assert L.sa_hip_csv_index_create(C.byref(h), b"/nonexistent/file.csv", b"name", 32, 0) == -1 and not h
    assert b"cannot open" in L.sa_hip_last_error()
    p = tmp_path / "t.csv"
    p.write_bytes(b"id,name\n1,abc\n")
    assert L.sa_hip_csv_index_create(C.byref(h), str(p).encode(), b"nope", 32, 0) == -1 and not h
    assert b"column not found" in L.sa_hip_last_error()
    assert L.sa_hip_csv_index_create(C.byref(h), str(p).encode(), b"name", 0, 0) == -1          # max_suffix_length must be >= 1
    assert L.sa_hip_csv_index_create(None, str(p).encode(), b"name", 32, 0) == -1
    n = C.c_uint32(0)
    assert L.sa_hip_get_matching_records_file(None, b"x", 4, None, C.byref(n)) == -1
    r = L.sa_hip_get_substring_positions_file(None, b"x")
    assert (r.first, r.second) == (0xFFFFFFFF, 0xFFFFFFFF)
    assert L.sa_hip_index_set_rows(None, None, 0) == -1
    assert L.sa_hip_index_query_rows(None, b"x", 1, 4, None, C.byref(n), None) == -1
    assert L.sa_hip_csv_index_num_rows(None) == 0 and L.sa_hip_csv_index_num_columns(None) == 0
    assert L.sa_hip_csv_index_column_name(None, 0) is None
    L.sa_hip_csv_index_destroy(None)   # no-op
    L.sa_hip_free_records(None, 3)     # no-op
    tab = (C.c_void_p * 2)()
    assert L.sa_hip_get_matching_records(None, None, b"x", 2, tab) == 0
