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
  golden_csv.npz     CSV mode through the reference's C functions (csv_golden below): hit counts and returned row ids
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


# ---- CSV mode: the reference's own C path (works at C level) -------------------------------------
def csv_golden():
    """golden_csv.npz: construct_truncated_suffix_array_from_csv_partitioned_mmap (engine.c:461-654, the body of ..._mmap_full)
    + get_substring_positions_file (engine.c:920-999) + get_matching_records_file (engine.c:1326-1390) of the compiled
    reference on (a) a tiny 4-row CSV with a quoted field and (b) a 10 000-row file of the synthetic config-5 generator
    (sa_hip_synth_csv, seed 7; the bytes are stored so the fixture is self-contained).  Per pattern: the number of
    hits (last - first + 1), the number of records returned and the sorted distinct `id`s of the returned rows (a
    returned record is identified by its text: the row minus its last and -- past the file's first page -- first character).

    What the reference does differently ON PURPOSE is not part of the fixture: it indexes the header row (patterns are
    chosen not to occur there: a hit in the first line makes rfc4180_seek_backward_newline return UINT32_MAX and the
    malloc that follows fails), it returns one record per HIT (a row that contains the pattern twice comes
    twice; ids are de-duplicated here) and a miss runs into undefined behaviour (engine.c:1347-1356 closes the file and
    carries on with first = UINT32_MAX), so only patterns that occur are asked.  `letters_only[i]` = the pattern has
    no byte below ',': the binary search compares FILE bytes (a field is followed by ',' there) while the array is
    sorted by the column text (a field is followed by newline), so for patterns with a space the search is not
    monotone and the reference may miss rows; `agrees_with_scan[i]` records whether its answer equals a plain scan of
    the column."""
    import ctypes as C
    import csv as _csv
    import io
    import tempfile
    from oracle.oracle import REF_SO, RefSuffixArrayStruct
    from suffixarray_amd import _capi
    L = C.CDLL(REF_SO)
    # the per-partition builder is called directly: its driver ..._mmap_full (engine.c:1454-1482) finds the column by
    # strcmp against header names that parse_csv_header copies WITHOUT a terminating NUL (engine.c:41-46), which only
    # works on freshly zeroed heap memory -- inside this process it picks a garbage column index and runs off the file
    L.init_suffix_array.restype = None
    L.init_suffix_array.argtypes = [C.c_void_p, C.c_uint32]
    L.construct_truncated_suffix_array_from_csv_partitioned_mmap.restype = None
    L.construct_truncated_suffix_array_from_csv_partitioned_mmap.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint16]
    L.get_matching_records_file.restype = None
    L.get_matching_records_file.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]

    class PairU32(C.Structure):
        _fields_ = [("first", C.c_uint32), ("second", C.c_uint32)]
    L.get_substring_positions_file.restype = PairU32
    L.get_substring_positions_file.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.free.argtypes = [C.c_void_p]

    tiny = (b"id,name,country\n1,Netflix Inc,US\n2,\"Netflix, Studios\",US\n3,Amazon,US\n4,Hulu LLC,US\n"
            b"5,amazonia netflix,BR\n")
    tmp = tempfile.mkdtemp()
    synth_path = os.path.join(tmp, "synth.csv")
    _capi.synth_csv(synth_path, 10_000, 7)
    files = {"tiny": (tiny, "name"), "synth10k": (open(synth_path, "rb").read(), "company_name")}
    arrays = {}
    rng = np.random.default_rng(4180)
    for name, (data, column) in files.items():
        path = os.path.join(tmp, name + ".csv")
        with open(path, "wb") as f:
            f.write(data)
        rows = list(_csv.reader(io.StringIO(data.decode())))
        header, body = rows[0], rows[1:]
        ci = header.index(column)
        fields = [r[ci].lower() for r in body]
        ids = [int(r[0]) for r in body]
        # a returned record is the row WITHOUT its last character (engine.c:1314) and, beyond the first 4 KiB page of the
        # file, also without its first one (rfc4180_seek_backward_newline, engine.c:1269: one short): rows are
        # identified by that text
        lines = data.decode().split("\n")[1:]
        lines = [l for l in lines if l]
        assert len(lines) == len(body)   # no newline inside a quoted field in these files
        by_text = {}
        for l, i in zip(lines, ids):
            for key in (l[:-1], l[1:-1]):
                assert by_text.get(key, i) == i, key
                by_text[key] = i
        # patterns: substrings of fields (they occur), not occurring in the header line
        pats = set()
        if name == "tiny":
            pats = {"netflix", "amazon", "hulu llc", "netflix, st", "inc", "zon", "flix"}
        else:
            while len(pats) < 300:
                f = fields[int(rng.integers(0, len(fields)))]
                m = int(rng.integers(2, 12))
                if len(f) < m:
                    continue
                o = int(rng.integers(0, len(f) - m + 1))
                pats.add(f[o:o + m])
        head_line = data.split(b"\n", 1)[0].decode().lower()
        pats = sorted(p for p in pats if p not in head_line and p.strip() == p and '"' not in p)
        st = RefSuffixArrayStruct()
        sa_struct = C.addressof(st)
        L.init_suffix_array(sa_struct, 32)                       # what init_suffix_array_index does per partition (engine.c:1446-1451)
        L.construct_truncated_suffix_array_from_csv_partitioned_mmap(path.encode(), ci, sa_struct, len(header))
        hit_counts, rec_counts, id_lists, letters, agrees = [], [], [], [], []
        for p in pats:
            fh = libc.fopen(path.encode(), b"r")
            r = L.get_substring_positions_file(fh, sa_struct, p.encode())
            libc.fclose(fh)
            scan = sorted(i for i, f in zip(ids, fields) if p in f)
            if r.first == 0xFFFFFFFF:
                # the pattern occurs, the reference's search does not find it (patterns with a byte below ',': see above);
                # recorded as such -- get_matching_records_file must not be reached with a miss
                assert not all(ord(c) >= 44 for c in p), (name, p)
                hit_counts.append(0); rec_counts.append(0); id_lists.append([]); letters.append(False); agrees.append(False)
                continue
            count = r.second - r.first + 1
            out = (C.c_void_p * count)()
            num = C.c_uint32(0)
            L.get_matching_records_file(path.encode(), sa_struct, p.encode(), count, out, C.byref(num))
            got = []
            for k in range(num.value):
                row = C.string_at(out[k]).decode()
                libc.free(out[k])
                got.append(by_text[row])
            hit_counts.append(count)
            rec_counts.append(num.value)
            id_lists.append(sorted(set(got)))
            letters.append(all(ord(c) >= 44 for c in p))
            agrees.append(sorted(set(got)) == scan)
        off = np.zeros(len(pats) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(x) for x in id_lists])
        arrays[f"csv__{name}"] = np.frombuffer(data, dtype=np.uint8)
        arrays[f"column__{name}"] = np.array([column])
        arrays[f"n__{name}"] = np.array([st.n], dtype=np.int64)   # characters the reference indexed (header row included)
        arrays[f"patterns__{name}"] = np.array(pats)
        arrays[f"hit_counts__{name}"] = np.array(hit_counts, dtype=np.int64)
        arrays[f"record_counts__{name}"] = np.array(rec_counts, dtype=np.int64)
        arrays[f"row_ids__{name}"] = np.array([i for l in id_lists for i in l], dtype=np.int64)
        arrays[f"row_ids_offsets__{name}"] = off
        arrays[f"letters_only__{name}"] = np.array(letters)
        arrays[f"agrees_with_scan__{name}"] = np.array(agrees)
        print(name, "patterns", len(pats), "letters-only", int(sum(letters)), "reference agrees with a scan:", int(sum(agrees)),
              "of which letters-only", int(sum(a and l for a, l in zip(agrees, letters))))
    arrays["names"] = np.array(list(files))
    arrays["max_suffix_length"] = np.array([32])
    np.savez_compressed(os.path.join(OUT, "golden_csv.npz"), **arrays)
    print("golden_csv.npz", os.path.getsize(os.path.join(OUT, "golden_csv.npz")))


if __name__ == "__main__":
    if "--csv-only" not in sys.argv:
        main()
    csv_golden()
