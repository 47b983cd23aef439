# cython: language_level=3
"""Cython binding of the C ABI (include/sa_hip.h): the reference's `cdef class SuffixArray`
(suffix_array/suffix_array.pyx:110-267, README.md:13-50) on top of the device index.

    from suffixarray_amd import SuffixArray          # = this class
    sa = SuffixArray(documents=docs, max_suffix_length=32)
    sa.query_records("the quick brown fox")
    sa = SuffixArray(csv_file="companies.csv", search_column="name", max_suffix_length=32)
    sa.query_records("netflix", k=10)

Everything below the Python objects is the C seam: construction (sa_hip_index_build / sa_hip_csv_index_create, the
replacement of construct_truncated_suffix_array[_from_csv_partitioned_mmap_full], pyx:176-207), the search and the record
retrieval (sa_hip_get_matching_records_file / sa_hip_index_query_rows, the replacement of get_matching_records_file,
pyx:224-232, with the reference's ownership rule: the callee mallocs the rows, this layer frees them, pyx:262-265).
Device calls run with the GIL released (the reference does the same around its engine calls, pyx:159-180,200-207).
Decisions where the reference snapshot cannot arbitrate: DESIGN.md section 9.
"""
from libc.stdint cimport uint8_t, uint32_t, uint64_t
from libc.stdlib cimport malloc, free
from libc.string cimport strlen, memchr
from cpython.unicode cimport PyUnicode_DecodeUTF8

import csv as _csv
import io as _io
import json as _json
import os as _os

import numpy as np


cdef extern from "sa_hip.h":
    ctypedef struct sa_hip_pair_u32:
        uint32_t first
        uint32_t second
    ctypedef struct sa_hip_index
    ctypedef struct sa_hip_csv_index
    int sa_hip_index_create(sa_hip_index** out, uint64_t n_max, int device) nogil
    void sa_hip_index_destroy(sa_hip_index* idx) nogil
    int sa_hip_index_build(sa_hip_index* idx, const uint8_t* T_host, uint64_t n, uint32_t max_suffix_length) nogil
    int sa_hip_index_load(sa_hip_index* idx, const uint8_t* T_host, const uint32_t* SA_host, uint64_t n, uint32_t max_suffix_length) nogil
    int sa_hip_query_batch(sa_hip_index* idx, const uint8_t* patterns, const uint64_t* offsets, uint64_t Q,
                           sa_hip_pair_u32* out) nogil
    int sa_hip_index_get_sa_u32(sa_hip_index* idx, uint32_t* out_host) nogil
    int sa_hip_index_get_text(sa_hip_index* idx, uint8_t* out_host) nogil
    uint64_t sa_hip_index_n(const sa_hip_index* idx) nogil
    int sa_hip_index_set_rows(sa_hip_index* idx, const uint64_t* row_text_starts, uint64_t num_rows) nogil
    int sa_hip_index_query_rows(sa_hip_index* idx, const uint8_t* pattern, uint64_t len, uint32_t k, uint64_t* row_ids,
                                uint32_t* num_rows, sa_hip_pair_u32* range) nogil
    int sa_hip_index_query_rows_batch(sa_hip_index* idx, const uint8_t* patterns, const uint64_t* offsets, uint64_t Q, uint32_t k,
                                      uint64_t* row_ids, uint32_t* counts, sa_hip_pair_u32* ranges) nogil
    int sa_hip_index_rows_for_range(sa_hip_index* idx, sa_hip_pair_u32 range, uint32_t k, uint64_t* row_ids, uint32_t* num_rows) nogil
    int sa_hip_csv_index_create(sa_hip_csv_index** out, const char* csv_file, const char* search_column,
                                uint32_t max_suffix_length, int device) nogil
    int sa_hip_csv_index_adopt(sa_hip_csv_index** out, const char* csv_file, const uint8_t* text, const uint32_t* SA, uint64_t n,
                               const uint64_t* row_text_starts, const uint64_t* row_file_offsets, uint64_t num_rows,
                               const char* column_names, uint32_t num_columns, uint32_t column_index,
                               uint32_t max_suffix_length, int device) nogil
    void sa_hip_csv_index_destroy(sa_hip_csv_index* c) nogil
    int sa_hip_csv_index_create_partitioned(sa_hip_csv_index*** out_parts, uint32_t* num_parts, const char* csv_file,
                                            const char* search_column, uint32_t max_suffix_length, int device,
                                            uint64_t partition_bytes) nogil
    void sa_hip_csv_index_free_parts(sa_hip_csv_index** parts) nogil
    sa_hip_index* sa_hip_csv_index_handle(sa_hip_csv_index* c) nogil
    int sa_hip_index_deep_keys(sa_hip_index* idx, int mode) nogil
    uint64_t sa_hip_csv_index_num_rows(const sa_hip_csv_index* c) nogil
    uint32_t sa_hip_csv_index_num_columns(const sa_hip_csv_index* c) nogil
    uint32_t sa_hip_csv_index_column_index(const sa_hip_csv_index* c) nogil
    const char* sa_hip_csv_index_column_name(const sa_hip_csv_index* c, uint32_t i) nogil
    int sa_hip_csv_index_row_tables(const sa_hip_csv_index* c, const uint64_t** row_text_starts, const uint64_t** row_file_offsets) nogil
    int sa_hip_get_matching_records_file(sa_hip_csv_index* c, const char* substring, uint32_t k, char** matching_records,
                                         uint32_t* num_matches) nogil
    int sa_hip_csv_index_copy_rows(sa_hip_csv_index* c, const uint64_t* row_ids, uint32_t n, char** records) nogil
    int sa_hip_get_matching_row_spans_file(sa_hip_csv_index* c, const char* substring, uint32_t k, const char** row_ptrs,
                                           uint32_t* row_lens, uint32_t* num_matches) nogil
    const char* sa_hip_last_error()


cdef bytes _LOWER = bytes((c + 32) if 65 <= c <= 90 else c for c in range(256))
FORMAT_VERSION = 3
ROWS_BUDGET = 1 << 24   # row ids per slice of query_records_batch (128 MB of ids on the host, 64 MB on the device)


def _map_array(path, dtype):
    """A saved array as a read-only memory map (an empty file as an empty array: mmap refuses length 0)."""
    if _os.path.getsize(path) == 0:
        return np.zeros(0, dtype=dtype)
    return np.memmap(path, dtype=dtype, mode="r")


cdef inline bytes ascii_lower(bytes b):
    # pyx:103-107 lowercase_string: only bytes 65..90 change
    return b.translate(_LOWER)


cdef _check(int rc):
    if rc != 0:
        msg = sa_hip_last_error().decode("utf-8", "replace")
        if rc == -1 and ("column not found" in msg or "cannot open" in msg or "empty CSV" in msg):
            raise ValueError(msg)
        raise RuntimeError("libsa_hip error %d: %s" % (rc, msg))


cdef list _shape_spans(list columns, const char** ptrs, const uint32_t* lens, uint32_t n):
    """_shape_rows over (pointer, length) spans of the mapped file: no per-row malloc / copy / strlen / free."""
    cdef list out = []
    cdef Py_ssize_t ncol = len(columns), c
    cdef uint32_t i
    cdef const char* p
    cdef const char* e
    cdef const char* q
    cdef dict d
    for i in range(n):
        p = ptrs[i]
        e = p + lens[i]
        if memchr(p, 34, lens[i]) != NULL:
            rec = next(_csv.reader(_io.StringIO(p[:lens[i]].decode("utf-8", "replace"))))
            out.append(dict(zip(columns, rec)))
            continue
        d = {}
        c = 0
        while c < ncol:
            q = <const char*>memchr(p, 44, e - p)
            if q == NULL:
                d[columns[c]] = PyUnicode_DecodeUTF8(p, e - p, "replace")
                break
            d[columns[c]] = PyUnicode_DecodeUTF8(p, q - p, "replace")
            p = q + 1
            c += 1
        out.append(d)
    return out


cdef list _shape_rows(list columns, char** recs, uint32_t n):
    """pyx:258-260 for malloc'ed rows, without intermediate objects: a row without a quote is cut at its commas in C and
    every field decoded straight into the dict (0.2 us per row instead of 0.7: at the reference's protocol -- 234 rows per
    query on average -- this was most of a query's latency); a quoted row goes through csv.reader."""
    cdef list out = []
    cdef Py_ssize_t ncol = len(columns), c
    cdef uint32_t i
    cdef const char* p
    cdef const char* e
    cdef const char* q
    cdef size_t L
    cdef dict d
    for i in range(n):
        p = recs[i]
        L = strlen(p)
        if memchr(p, 34, L) != NULL:
            rec = next(_csv.reader(_io.StringIO(p[:L].decode("utf-8", "replace"))))
            out.append(dict(zip(columns, rec)))
            continue
        e = p + L
        d = {}
        c = 0
        while c < ncol:
            q = <const char*>memchr(p, 44, e - p)
            if q == NULL:
                d[columns[c]] = PyUnicode_DecodeUTF8(p, e - p, "replace")
                break
            d[columns[c]] = PyUnicode_DecodeUTF8(p, q - p, "replace")
            p = q + 1
            c += 1
        out.append(d)
    return out


cdef class SuffixArray:
    cdef sa_hip_index* _idx          # documents mode: owned; CSV mode: borrowed from _csv
    cdef sa_hip_csv_index* _csv
    cdef uint32_t _L
    cdef int device
    cdef str _mode
    cdef list _documents
    cdef object _row_starts
    cdef public list columns
    cdef public str csv_filename
    cdef list _parts                 # documents beyond one index's 2^32 - 2 bytes: one SuffixArray per partition
    cdef uint64_t _partition_bytes

    def __cinit__(self):
        self._idx = NULL
        self._csv = NULL

    def __init__(self, documents=None, csv_file=None, search_column=None, max_suffix_length: int = 64,
                 device: int = 0, partition_bytes=None):
        """partition_bytes: documents whose joined text exceeds it are indexed as several partitions of whole documents,
        each a device index of its own, queried one after the other until k records are found -- the reference's scheme
        for large inputs (2 GiB partitions, suffix_array.pyx:148-180, 221-247), cut at document boundaries here so that no
        match is lost at a cut.  Default: 4e9 bytes (one index holds up to 2^32 - 2)."""
        if max_suffix_length is None or int(max_suffix_length) < 1:
            raise ValueError("max_suffix_length must be >= 1")
        self._L = <uint32_t>int(max_suffix_length)
        self.device = int(device)
        self._mode = ""
        self._parts = None
        self._partition_bytes = 4_000_000_000 if partition_bytes is None else max(1, min(int(partition_bytes), 4_000_000_000))
        if documents is not None and csv_file is not None:
            raise ValueError("pass either documents= or csv_file=, not both")
        if documents is not None:
            self.construct_truncated_suffix_array_documents(documents)
        elif csv_file is not None:
            if search_column is None:
                raise ValueError("search_column is required with csv_file")
            self.construct_truncated_suffix_array_from_csv(csv_file, search_column)

    def __dealloc__(self):
        self._release()

    cdef _release(self):
        if self._parts is not None:
            for p in self._parts:
                p.close()
            self._parts = None
        if self._csv != NULL:
            sa_hip_csv_index_destroy(self._csv)   # owns the device index
            self._csv = NULL
            self._idx = NULL
        if self._idx != NULL:
            sa_hip_index_destroy(self._idx)
            self._idx = NULL

    def close(self):
        self._release()

    @property
    def max_suffix_length(self):
        return self._L

    @property
    def handle(self):
        """The sa_hip_index* underneath, as an integer (0 = none)."""
        return <size_t>self._idx

    @property
    def _index(self):
        """The device index as the ctypes wrapper of the handle API (statistics, verification, batched device calls);
        it does not own the handle."""
        from suffixarray_amd import _capi
        if self._idx == NULL:
            return None
        return _capi.DeviceIndex.from_handle(<size_t>self._idx, self)

    # -- construction -------------------------------------------------------------------------------------------------
    cdef _build_documents(self, object text, object row_starts, object sa):
        """text: the joined lower-cased documents (bytes, or any read-only byte buffer: load() hands over a memory-mapped file);
        sa: None = build, else a saved suffix array to adopt"""
        cdef uint64_t n = len(text)
        cdef const uint8_t[::1] tview = text if n else b"\0"
        cdef const uint8_t* p = &tview[0]
        cdef uint32_t L = self._L
        cdef int rc
        cdef const uint64_t[::1] rs
        cdef const uint32_t[::1] sv
        if n > 0xFFFFFFFE:
            raise ValueError("text exceeds 2^32 - 2 bytes (one index per device)")
        self._release()
        _check(sa_hip_index_create(&self._idx, n if n > 0 else 1, self.device))
        if sa is None:
            with nogil:
                rc = sa_hip_index_build(self._idx, p if n else NULL, n, L)
        else:
            sv = sa
            with nogil:
                rc = sa_hip_index_load(self._idx, p if n else NULL, &sv[0] if n else NULL, n, L)
        _check(rc)
        self._row_starts = np.ascontiguousarray(row_starts, dtype=np.uint64)
        rs = self._row_starts
        _check(sa_hip_index_set_rows(self._idx, &rs[0] if rs.shape[0] else NULL, rs.shape[0]))
        self._deep_keys()

    cdef _deep_keys(self):
        # second-level keys (sa_hip_index_deep_keys): this class serves names and documents -- patterns longer than the key, shared
        # prefixes; a gather over the tied slots at construction (16 ms at 50M rows) takes a tenth off every query_records
        # (the reference's protocol: mean 136 -> 121 us, median 48 -> 43 us); 0 = the index has no use for them (narrow keys)
        cdef int rc
        with nogil:
            rc = sa_hip_index_deep_keys(self._idx, 2)
        if rc < 0:
            _check(rc)

    def construct_truncated_suffix_array_documents(self, documents):
        """pyx:129-180: text = '\\n'.join(documents), lower-cased, one truncated SA over it."""
        if not isinstance(documents, list):
            try:
                documents = list(documents)
            except Exception:
                raise ValueError("Documents must be a list of strings")
        self._documents = documents
        # ASCII documents without embedded newlines (the usual case): ONE join + encode, the row starts from the newline positions
        # -- per-document encode() and len() calls were 0.4 s of a 0.5 s construction for 5M short documents (20 ms on the device)
        try:
            joined = "\n".join(documents) if documents else None
        except TypeError:
            joined = None
        if joined is not None and len(joined) <= self._partition_bytes and joined.isascii() and joined.count("\n") == len(documents) - 1:
            raw = joined.encode("ascii")
            nl = np.flatnonzero(np.frombuffer(raw, dtype=np.uint8) == 10)
            starts = np.concatenate([np.zeros(1, np.int64), nl + 1]).astype(np.uint64)
            self._build_documents(ascii_lower(raw), starts, None)
            self._mode = "documents"
            return
        encoded = [d.encode("utf-8") for d in documents]
        lens = np.fromiter((len(e) for e in encoded), dtype=np.int64, count=len(encoded))
        if len(encoded) and int(lens.sum()) + len(encoded) - 1 > self._partition_bytes:
            # several partitions of whole documents (pyx:148-180 cuts the TEXT every 2 GiB and loses matches at the cuts)
            self._release()
            self._parts = []
            lo = 0
            acc = 0
            for i in range(len(encoded)):
                add = int(lens[i]) + (1 if i > lo else 0)
                if i > lo and acc + add > self._partition_bytes:
                    self._parts.append(SuffixArray(documents=documents[lo:i], max_suffix_length=self._L, device=self.device,
                                                   partition_bytes=4_000_000_000))
                    lo, acc, add = i, 0, int(lens[i])
                if add > 4_000_000_000:
                    raise ValueError("a single document exceeds 2^32 - 2 bytes (one index per partition)")
                acc += add
            self._parts.append(SuffixArray(documents=documents[lo:], max_suffix_length=self._L, device=self.device, partition_bytes=4_000_000_000))
            self._mode = "partitioned"
            return
        starts = np.concatenate([[0], np.cumsum(lens + 1)[:-1]]).astype(np.uint64) if len(encoded) else np.zeros(0, np.uint64)
        self._build_documents(ascii_lower(b"\n".join(encoded)), starts, None)
        self._mode = "documents"

    def construct_truncated_suffix_array_from_csv(self, str filename, str search_column):
        """pyx:183-207 / engine.c:1454-1482: index one column of a CSV file (RFC-4180 quoting) -- the whole of it is one
        call into the C seam (native extractor, device build, row tables, file mapping)."""
        cdef bytes fn = _os.fsencode(filename)
        cdef bytes col = search_column.encode("utf-8")
        cdef const char* fnp = fn
        cdef const char* colp = col
        cdef int rc
        cdef uint32_t L = self._L
        cdef int device = self.device
        cdef sa_hip_csv_index** parts = NULL
        cdef uint32_t nparts = 0, i
        cdef uint64_t pb = self._partition_bytes
        cdef SuffixArray part
        self._release()
        with nogil:
            rc = sa_hip_csv_index_create_partitioned(&parts, &nparts, fnp, colp, L, device, pb)
        _check(rc)
        try:
            if nparts == 1:
                self._csv = parts[0]
                parts[0] = NULL
                self._adopt_csv(filename)
                return
            # a column beyond partition_bytes: one index per run of whole rows (engine.c:1437-1481 cuts the file every 2 GiB),
            # answered one after the other (pyx:221-247); each part is a CSV-mode SuffixArray of its own
            self._parts = []
            for i in range(nparts):
                part = SuffixArray.__new__(SuffixArray)
                part._L = L
                part.device = device
                part._parts = None
                part._partition_bytes = pb
                part._csv = parts[i]
                parts[i] = NULL
                part._adopt_csv(filename)
                self._parts.append(part)
            self.csv_filename = filename
            self.columns = list(self._parts[0].columns)
            self._mode = "partitioned"
        finally:
            for i in range(nparts):
                if parts[i] != NULL:
                    sa_hip_csv_index_destroy(parts[i])
            sa_hip_csv_index_free_parts(parts)

    cdef _adopt_csv(self, str filename):
        cdef uint32_t i
        self._idx = sa_hip_csv_index_handle(self._csv)
        self._deep_keys()
        self.csv_filename = filename
        self.columns = [sa_hip_csv_index_column_name(self._csv, i).decode("utf-8") for i in range(sa_hip_csv_index_num_columns(self._csv))]
        self._mode = "csv"

    # -- query --------------------------------------------------------------------------------------------------------
    def query_ranges(self, substrings):
        """Batched get_substring_positions (engine.c:869-918 per element): structured (first, second)."""
        if self._mode == "partitioned":
            raise RuntimeError("a partitioned index has one suffix array per partition: use .partitions[i].query_ranges")
        if self._idx == NULL:
            raise RuntimeError("index not built")
        pats = [ascii_lower(s.encode("utf-8")) if isinstance(s, str) else ascii_lower(bytes(s)) for s in substrings]
        cdef uint64_t Q = len(pats)
        out = np.zeros(max(Q, 1), dtype=np.dtype([("first", "<u4"), ("second", "<u4")]))
        if Q == 0:
            return out[:0]
        off = np.zeros(Q + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(p) for p in pats], dtype=np.uint64)
        cdef bytes buf = b"".join(pats) + b"\0"
        cdef const uint8_t* bp = <const uint8_t*>(<const char*>buf)
        cdef uint64_t[::1] offv = off
        cdef unsigned char[::1] outv = out.view(np.uint8)
        cdef int rc
        with nogil:
            rc = sa_hip_query_batch(self._idx, bp, &offv[0], Q, <sa_hip_pair_u32*>&outv[0])
        _check(rc)
        return out[:Q]

    cdef list _csv_rows(self, const uint64_t* row_ids, uint32_t n):
        """rows of the file by id -> list of dicts (the rows are malloc'ed by the callee, freed here: pyx:262-265)"""
        cdef char** recs = <char**>malloc(max(n, 1) * sizeof(char*))
        cdef int rc
        cdef uint32_t i
        if recs == NULL:
            raise MemoryError()
        with nogil:
            rc = sa_hip_csv_index_copy_rows(self._csv, row_ids, n, recs)
        if rc != 0:
            free(recs)
            _check(rc)
        try:
            return _shape_rows(self.columns, recs, n)
        finally:
            for i in range(n):
                free(recs[i])
            free(recs)

    def query_records(self, substring: str, k: int = 1000):
        """pyx:209-267: records containing `substring` (case-insensitive ASCII), at most k."""
        if substring == "" or k <= 0:
            return []
        if self._mode == "partitioned":
            # pyx:221-247: the partitions one after the other until k records are found
            out = []
            for p in self._parts:
                out += p.query_records(substring, k - len(out))
                if len(out) >= k:
                    break
            return out
        if self._idx == NULL:
            raise RuntimeError("index not built")
        cdef bytes pat = ascii_lower(substring.encode("utf-8") if isinstance(substring, str) else bytes(substring))
        cdef const char* pp = pat
        # never more rows than the index has: k = 10**9 ("all") must not size any buffer
        cdef uint64_t nrows = sa_hip_csv_index_num_rows(self._csv) if self._mode == "csv" else len(self._row_starts)
        cdef uint32_t kk = <uint32_t>min(int(k), max(nrows, 1), 0x7FFFFFFF)
        cdef uint32_t n = 0
        cdef uint64_t plen = len(pat)
        cdef int rc
        cdef uint32_t i
        cdef char** recs
        cdef const char** spans
        cdef uint32_t* lens
        cdef uint64_t[::1] rv
        if self._mode == "csv" and b"\0" not in pat:
            # the reference's own call (pyx:224-232 -> get_matching_records_file) in its zero-copy form: the dicts are built straight
            # from the rows' bytes in the mapped file (sa_hip_get_matching_records_file mallocs every row for a C caller to free)
            spans = <const char**>malloc(kk * sizeof(char*))
            lens = <uint32_t*>malloc(kk * sizeof(uint32_t))
            if spans == NULL or lens == NULL:
                free(spans); free(lens)
                raise MemoryError()
            with nogil:
                rc = sa_hip_get_matching_row_spans_file(self._csv, pp, kk, spans, lens, &n)
            try:
                _check(rc)
                return _shape_spans(self.columns, spans, lens, n)
            finally:
                free(spans)
                free(lens)
        # (a malloc'ed id buffer and a C loop: numpy array + typed memoryview + per-element int() were a third of a documents-mode call)
        cdef uint64_t* ids = <uint64_t*>malloc(kk * sizeof(uint64_t))
        cdef list docs = self._documents
        if ids == NULL:
            raise MemoryError()
        try:
            with nogil:
                rc = sa_hip_index_query_rows(self._idx, <const uint8_t*>pp, plen, kk, ids, &n, NULL)
            _check(rc)
            if self._mode == "csv":
                return self._csv_rows(ids, n)
            return [docs[ids[i]] for i in range(n)]
        finally:
            free(ids)

    def query_records_batch(self, substrings, k: int = 1000):
        """The batched form: ONE launch finds every range, ONE more maps every hit of every range to its row and
        de-duplicates per query on the device (sa_hip_index_query_rows_batch); the host only copies the rows out."""
        live = [i for i, s in enumerate(substrings) if s != ""]
        res = [[] for _ in substrings]
        if not live or k <= 0:
            return res
        if self._mode == "partitioned":
            for p in self._parts:     # one batched call per partition
                for i, r in enumerate(p.query_records_batch(substrings, k)):
                    if len(res[i]) < k:
                        res[i] += r[:k - len(res[i])]
            return res
        if self._idx == NULL:
            raise RuntimeError("index not built")
        pats = [ascii_lower(substrings[i].encode("utf-8")) if isinstance(substrings[i], str) else ascii_lower(bytes(substrings[i])) for i in live]
        cdef uint64_t Q = len(pats)
        cdef uint64_t nrows = sa_hip_csv_index_num_rows(self._csv) if self._mode == "csv" else len(self._row_starts)
        cdef uint32_t kk = <uint32_t>min(int(k), max(nrows, 1), 0x7FFFFFFF)
        off = np.zeros(Q + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(p) for p in pats], dtype=np.uint64)
        cdef bytes buf = b"".join(pats) + b"\0"
        cdef const uint8_t* bp = <const uint8_t*>(<const char*>buf)
        cdef uint64_t[::1] offv = off
        # The library sizes host and device buffers as Q x k row ids: the batch goes through in slices of at most
        # ROWS_BUDGET ids (k = 1000, the reference protocol's default, times 100 000 patterns would be 0.8 GB of ids here and
        # 0.4 GB on the device for results that mostly hold a handful of rows), the buffers reused from slice to slice.
        cdef uint64_t per = max(<uint64_t>1, <uint64_t>ROWS_BUDGET // max(<uint64_t>kk, <uint64_t>1))
        if per > Q:
            per = Q
        rows = np.empty((per, kk), dtype=np.uint64)
        counts = np.zeros(per, dtype=np.uint32)
        soff = np.zeros(per + 1, dtype=np.uint64)
        cdef uint64_t[:, ::1] rv = rows
        cdef uint32_t[::1] cv = counts
        cdef uint64_t[::1] sov = soff
        cdef int rc
        cdef uint64_t j, lo = 0, qs, base
        while lo < Q:
            qs = min(per, Q - lo)
            base = offv[lo]
            for j in range(qs + 1):
                sov[j] = offv[lo + j] - base
            with nogil:
                rc = sa_hip_index_query_rows_batch(self._idx, bp + base, &sov[0], qs, kk, &rv[0, 0], &cv[0], NULL)
            _check(rc)
            for j in range(qs):
                if self._mode == "csv":
                    res[live[lo + j]] = self._csv_rows(&rv[j, 0], cv[j])
                else:
                    res[live[lo + j]] = [self._documents[int(r)] for r in rows[j, :cv[j]]]
            lo += qs
        return res

    # -- persistence (SURVEY.md 8(f)-3; the reference's save / load is half-built: engine.c:1098-1165, commented-out
    #    pyx:310-423).  Versioned directory: meta.json + raw little-endian arrays; load() adopts them, no rebuild. -------
    @property
    def partitions(self):
        """The per-partition indexes of a partitioned (documents or CSV) index (else [self])."""
        return list(self._parts) if self._parts is not None else [self]

    def save(self, directory: str):
        if self._mode == "partitioned":
            _os.makedirs(directory, exist_ok=True)
            for i, p in enumerate(self._parts):
                p.save(_os.path.join(directory, "part%04d" % i))
            with open(_os.path.join(directory, "meta.json"), "w") as f:
                _json.dump({"format": "suffixarray_amd", "version": FORMAT_VERSION, "mode": "partitioned", "parts": len(self._parts),
                            "max_suffix_length": int(self._L)}, f)
            return
        if self._idx == NULL:
            raise RuntimeError("index not built")
        _os.makedirs(directory, exist_ok=True)
        cdef uint64_t n = sa_hip_index_n(self._idx)
        sa = np.empty(max(n, 1), dtype=np.uint32)
        text = np.empty(max(n, 1), dtype=np.uint8)
        cdef uint32_t[::1] sv = sa
        cdef uint8_t[::1] tv = text
        cdef int rc
        cdef const uint64_t* rts = NULL
        cdef const uint64_t* rfo = NULL
        cdef uint64_t rows
        with nogil:
            rc = sa_hip_index_get_sa_u32(self._idx, &sv[0])
            if rc == 0:
                rc = sa_hip_index_get_text(self._idx, &tv[0])
        _check(rc)
        sa[:n].tofile(_os.path.join(directory, "sa.u32"))
        text[:n].tofile(_os.path.join(directory, "text.u8"))
        meta = {"format": "suffixarray_amd", "version": FORMAT_VERSION, "mode": self._mode, "n": int(n),
                "max_suffix_length": int(self._L), "columns": self.columns}
        if self._mode == "documents":
            np.asarray(self._row_starts, dtype=np.uint64).tofile(_os.path.join(directory, "row_starts.u64"))
            with open(_os.path.join(directory, "documents.json"), "w") as f:
                _json.dump(self._documents, f)
        else:
            _check(sa_hip_csv_index_row_tables(self._csv, &rts, &rfo))
            rows = sa_hip_csv_index_num_rows(self._csv)
            np.asarray(<uint64_t[:max(rows, 1)]>rts)[:rows].tofile(_os.path.join(directory, "row_starts.u64"))
            np.asarray(<uint64_t[:rows + 1]>rfo).tofile(_os.path.join(directory, "row_file_offsets.u64"))
            meta["csv_filename"] = _os.path.abspath(self.csv_filename)
            meta["column_index"] = int(sa_hip_csv_index_column_index(self._csv))
            # the identity of the file the row offsets point into: load() refuses a file that has changed
            st = _os.stat(self.csv_filename)
            meta["csv_size"] = int(st.st_size)
            meta["csv_mtime_ns"] = int(st.st_mtime_ns)
        with open(_os.path.join(directory, "meta.json"), "w") as f:
            _json.dump(meta, f)

    @classmethod
    def load(cls, directory: str, device: int = 0):
        """Re-open a saved index: uploads text + SA (sa_hip_index_load / sa_hip_csv_index_adopt), no construction."""
        with open(_os.path.join(directory, "meta.json")) as f:
            meta = _json.load(f)
        # version 2 (round 2) is version 3 without the CSV file's identity (csv_size / csv_mtime_ns) and without partitions
        if meta.get("format") != "suffixarray_amd" or meta.get("version") not in (2, FORMAT_VERSION):
            raise ValueError("not a suffixarray_amd index of a supported version")
        cdef SuffixArray self = cls(max_suffix_length=meta["max_suffix_length"], device=device)
        if meta.get("mode") == "partitioned":
            self._parts = [cls.load(_os.path.join(directory, "part%04d" % i), device) for i in range(int(meta["parts"]))]
            if self._parts and (<SuffixArray>self._parts[0])._mode == "csv":
                self.columns = list((<SuffixArray>self._parts[0]).columns)
                self.csv_filename = (<SuffixArray>self._parts[0]).csv_filename
            else:
                self._documents = [d for p in self._parts for d in (<SuffixArray>p)._documents]
            self._mode = "partitioned"
            return self
        # memory-mapped, read-only: the arrays stream from the page cache into the upload, so an index larger than the host's
        # free memory re-opens (np.fromfile held text + SA + row tables in anonymous memory: 5 n + 16 rows bytes)
        text = _map_array(_os.path.join(directory, "text.u8"), np.uint8)
        sa = _map_array(_os.path.join(directory, "sa.u32"), np.uint32)
        if text.size != meta["n"] or sa.size != meta["n"]:
            raise ValueError("index files are truncated")
        starts = _map_array(_os.path.join(directory, "row_starts.u64"), np.uint64)
        if starts.size and (starts[0] != 0 or starts[-1] > meta["n"] or np.any(starts[1:] < starts[:-1])):
            raise ValueError("index files are corrupt (row table)")
        self.columns = meta["columns"]
        if meta["mode"] == "documents":
            with open(_os.path.join(directory, "documents.json")) as f:
                self._documents = _json.load(f)
            if len(self._documents) != starts.size:
                raise ValueError("index files are truncated")
            self._build_documents(text, starts, sa if sa.size else np.zeros(1, np.uint32))
            self._mode = "documents"
            return self
        offs = _map_array(_os.path.join(directory, "row_file_offsets.u64"), np.uint64)
        if offs.size != starts.size + 1:
            raise ValueError("index files are truncated")
        try:
            st = _os.stat(meta["csv_filename"])
        except OSError:
            raise ValueError("the CSV file of this index is gone: " + meta["csv_filename"])
        if "csv_size" not in meta:
            import warnings
            warnings.warn("index saved by format version 2: the CSV file's identity was not recorded and cannot be checked")
        elif int(st.st_size) != meta.get("csv_size") or int(st.st_mtime_ns) != meta.get("csv_mtime_ns"):
            raise ValueError("the CSV file has changed since the index was saved (size / modification time): rebuild the index")
        cdef bytes fn = _os.fsencode(meta["csv_filename"])
        cdef bytes names = b"".join(c.encode("utf-8") + b"\0" for c in meta["columns"])
        cdef const char* fnp = fn
        cdef const char* namesp = names
        cdef const uint8_t[::1] tv = text if text.size else np.zeros(1, np.uint8)
        cdef const uint32_t[::1] sv = sa if sa.size else np.zeros(1, np.uint32)
        cdef const uint64_t[::1] stv = starts if starts.size else np.zeros(1, np.uint64)
        cdef const uint64_t[::1] ofv = offs
        cdef uint64_t n = text.size, rows = starts.size
        cdef uint32_t ncols = len(meta["columns"]), ci = meta["column_index"], L = self._L
        cdef int rc, dev = self.device
        with nogil:
            rc = sa_hip_csv_index_adopt(&self._csv, fnp, &tv[0], &sv[0], n, &stv[0], &ofv[0], rows, namesp, ncols, ci, L, dev)
        _check(rc)
        self._adopt_csv(meta["csv_filename"])
        return self
