# cython: language_level=3
"""Cython binding of the C ABI (include/sa_hip.h): the reference's `cdef class SuffixArray`
(suffix_array/suffix_array.pyx:110-267, README.md:13-50) on top of the device index.

    from suffixarray_amd.suffix_array import SuffixArray
    sa = SuffixArray(documents=docs, max_suffix_length=32)
    sa.query_records("the quick brown fox")

Device calls run with the GIL released (the reference does the same around its engine calls,
pyx:159-180,200-207).  Construction / record shaping decisions: DESIGN.md section 9.
"""
from libc.stdint cimport uint8_t, uint32_t, uint64_t
from libc.stdlib cimport malloc, free

import csv as _csv
import io as _io

import numpy as np


cdef extern from "sa_hip.h":
    ctypedef struct sa_hip_pair_u32:
        uint32_t first
        uint32_t second
    ctypedef struct sa_hip_index
    int sa_hip_index_create(sa_hip_index** out, uint64_t n_max, int device) nogil
    void sa_hip_index_destroy(sa_hip_index* idx) nogil
    int sa_hip_index_build(sa_hip_index* idx, const uint8_t* T_host, uint64_t n, uint32_t max_suffix_length) nogil
    int sa_hip_query_batch(sa_hip_index* idx, const uint8_t* patterns, const uint64_t* offsets, uint64_t Q,
                           sa_hip_pair_u32* out) nogil
    int sa_hip_index_get_sa_range(sa_hip_index* idx, uint64_t first, uint64_t count, uint32_t* out_host) nogil
    int sa_hip_index_query_hits(sa_hip_index* idx, const uint8_t* pattern, uint64_t len, uint32_t max_hits,
                                sa_hip_pair_u32* range, uint32_t* hits, uint32_t* nhits) nogil
    uint64_t sa_hip_index_n(const sa_hip_index* idx) nogil
    const char* sa_hip_last_error()


cdef bytes _LOWER = bytes((c + 32) if 65 <= c <= 90 else c for c in range(256))


cdef inline bytes ascii_lower(bytes b):
    # pyx:103-107 lowercase_string: only bytes 65..90 change
    return b.translate(_LOWER)


cdef _check(int rc):
    if rc != 0:
        raise RuntimeError("libsa_hip error %d: %s" % (rc, sa_hip_last_error().decode("utf-8", "replace")))


cdef class SuffixArray:
    cdef sa_hip_index* _idx
    cdef uint32_t max_suffix_length
    cdef int device
    cdef str _mode
    cdef list _documents
    cdef object _row_starts
    cdef object _row_file_offsets
    cdef object _csv_mm
    cdef object _csv_fh
    cdef public list columns
    cdef public str csv_filename

    def __cinit__(self):
        self._idx = NULL

    def __init__(self, documents=None, csv_file=None, search_column=None, max_suffix_length: int = 64,
                 device: int = 0):
        if max_suffix_length is None or int(max_suffix_length) < 1:
            raise ValueError("max_suffix_length must be >= 1")
        self.max_suffix_length = <uint32_t>int(max_suffix_length)
        self.device = int(device)
        self._mode = ""
        if documents is not None and csv_file is not None:
            raise ValueError("pass either documents= or csv_file=, not both")
        if documents is not None:
            self.construct_truncated_suffix_array_documents(documents)
        elif csv_file is not None:
            if search_column is None:
                raise ValueError("search_column is required with csv_file")
            self.construct_truncated_suffix_array_from_csv(csv_file, search_column)

    def __dealloc__(self):
        if self._idx != NULL:
            sa_hip_index_destroy(self._idx)
            self._idx = NULL

    def close(self):
        if self._idx != NULL:
            sa_hip_index_destroy(self._idx)
            self._idx = NULL
        if self._csv_mm is not None:
            self._csv_mm.close()
            self._csv_fh.close()
            self._csv_mm = None

    cdef _set_text(self, text):
        """text: bytes or a C-contiguous uint8 array (the CSV extractor's buffer, not copied)"""
        cdef const uint8_t[::1] mv = text
        cdef uint64_t n = mv.shape[0]
        cdef const uint8_t* p = &mv[0] if n else NULL
        cdef int rc
        cdef uint32_t L = self.max_suffix_length
        if n > 0xFFFFFFFE:
            raise ValueError("text exceeds 2^32 - 2 bytes (one index per device)")
        if self._idx != NULL:
            sa_hip_index_destroy(self._idx)
            self._idx = NULL
        _check(sa_hip_index_create(&self._idx, n if n > 0 else 1, self.device))
        with nogil:
            rc = sa_hip_index_build(self._idx, p, n, L)
        _check(rc)

    def construct_truncated_suffix_array_documents(self, documents):
        """pyx:129-180: text = '\\n'.join(documents), lower-cased, one truncated SA over it."""
        if not isinstance(documents, list):
            try:
                documents = list(documents)
            except Exception:
                raise ValueError("Documents must be a list of strings")
        self._documents = documents
        encoded = [d.encode("utf-8") for d in documents]
        lens = np.fromiter((len(e) for e in encoded), dtype=np.int64, count=len(encoded))
        if len(encoded):
            self._row_starts = np.concatenate([[0], np.cumsum(lens + 1)[:-1]]).astype(np.int64)
        else:
            self._row_starts = np.zeros(0, np.int64)
        self._set_text(ascii_lower(b"\n".join(encoded)))
        self._mode = "documents"

    def construct_truncated_suffix_array_from_csv(self, str filename, str search_column):
        """pyx:183-207 / engine.c:461-654: index one column of a CSV file (RFC-4180 quoting)."""
        from suffixarray_amd.csv_ingest import extract_column
        self.csv_filename = filename
        col = extract_column(filename, search_column)
        self.columns = col.columns
        self._row_starts = col.text_row_starts
        self._row_file_offsets = col.row_file_offsets
        self._set_text(col.text_array)   # a view of the extractor's own buffer: no copy
        self._mode = "csv"

    def query_ranges(self, substrings):
        """Batched get_substring_positions (engine.c:869-918 per element): structured (first, second)."""
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

    cdef _rows_for_range(self, uint32_t first, uint32_t second, int k, first_hits=None):
        # first_hits: SA[first .. first + len) already fetched by sa_hip_index_query_hits
        if first == 0xFFFFFFFF or ((second - first + 1) & 0xFFFFFFFF) == 0:
            return []
        cdef uint64_t pos = first
        cdef uint64_t end = <uint64_t>second + 1
        cdef uint64_t take
        cdef uint64_t slab = max(4 * k, 1024)
        cdef int rc
        cdef uint32_t[::1] hv
        rows = []
        seen = set()
        hits = np.empty(slab, dtype=np.uint32)
        hv = hits
        while pos < end and len(rows) < k:
            take = min(slab, end - pos)
            if first_hits is not None and pos == first and len(first_hits):
                take = min(take, <uint64_t>len(first_hits))
                hits[:take] = first_hits[:take]
            else:
                with nogil:
                    rc = sa_hip_index_get_sa_range(self._idx, pos, take, &hv[0])
                _check(rc)
            ids = np.searchsorted(self._row_starts, hits[:take].astype(np.int64), side="right") - 1
            _, first_at = np.unique(ids, return_index=True)   # distinct rows in order of first appearance
            for r in ids[np.sort(first_at)].tolist():
                if r not in seen:
                    seen.add(r)
                    rows.append(r)
                    if len(rows) == k:
                        break
            pos += take
        return rows

    cdef _materialise(self, list rows):
        if self._mode == "documents":
            return [self._documents[r] for r in rows]
        # the file is mapped once (the reference re-opens it and does one fseek + fread per row,
        # engine.c:1334-1390); a row without a quote character is split directly
        if self._csv_mm is None:
            import mmap
            self._csv_fh = open(self.csv_filename, "rb")
            self._csv_mm = mmap.mmap(self._csv_fh.fileno(), 0, access=mmap.ACCESS_READ)
        mm = self._csv_mm
        off = self._row_file_offsets
        out = []
        for r in rows:
            raw = mm[int(off[r]):int(off[r + 1])]
            if b'"' in raw:
                rec = next(_csv.reader(_io.StringIO(raw.decode("utf-8", "replace"))))
            else:
                rec = raw.decode("utf-8", "replace").rstrip("\r\n").split(",")
            out.append(dict(zip(self.columns, rec)))
        return out

    def query_records(self, substring: str, k: int = 1000):
        """pyx:209-267: records containing `substring` (case-insensitive ASCII), at most k."""
        if substring == "":
            return []
        # one call fetches the range and the first hits (no copy calls, one synchronisation)
        cdef bytes pat = ascii_lower(substring.encode("utf-8"))
        cdef const uint8_t* pp = <const uint8_t*>(<const char*>pat)
        cdef uint64_t plen = len(pat)
        cdef uint32_t cap = min(max(4 * k, 1024), 4096)
        cdef uint32_t nh = 0
        cdef sa_hip_pair_u32 rng
        cdef int rc
        fh = np.empty(cap, dtype=np.uint32)
        cdef uint32_t[::1] fv = fh
        with nogil:
            rc = sa_hip_index_query_hits(self._idx, pp, plen, cap, &rng, &fv[0], &nh)
        _check(rc)
        return self._materialise(self._rows_for_range(rng.first, rng.second, k, fh[:nh]))

    def query_records_batch(self, substrings, k: int = 1000):
        live = [i for i, s in enumerate(substrings) if s != ""]
        res = [[] for _ in substrings]
        if not live:
            return res
        ranges = self.query_ranges([substrings[i] for i in live])
        for j, i in enumerate(live):
            rows = self._rows_for_range(int(ranges[j]["first"]), int(ranges[j]["second"]), k)
            res[i] = self._materialise(rows)
        return res
