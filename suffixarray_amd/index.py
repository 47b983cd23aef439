"""Host-side mirror of the reference's Python class (suffix_array/suffix_array.pyx:110-267,
README.md:13-50) on top of the C ABI.  Same names, argument meaning and defaults:

    SuffixArray(documents=[...], max_suffix_length=32).query_records("the quick brown fox")
    SuffixArray(csv_file=..., search_column=..., max_suffix_length=32).query_records("netflix")

plus the batched entry point the device path exists for: query_records_batch / query_ranges.
Decisions where the reference snapshot cannot arbitrate (SURVEY.md 0.3, 7 "hard parts") are
listed in DESIGN.md: documents are returned in their original case, one record per matching
document (no duplicates), CSV header row is not indexed.
"""
import csv as _csv
import sys
import time
import io
import os

import numpy as np

from . import _capi


def _ascii_lower(b: bytes) -> bytes:
    """pyx:103-107 lowercase_string: only bytes 65..90 are changed."""
    return b.translate(_LOWER)


_LOWER = bytes((c + 32) if 65 <= c <= 90 else c for c in range(256))


def _warm_device(device):
    """First HIP use of the process on `device`: a one-byte index is created and dropped."""
    try:
        _capi.DeviceIndex(1, device).close()
    except Exception:
        pass   # the real create reports the error


class SuffixArray:
    def __init__(self, documents=None, csv_file=None, search_column=None, max_suffix_length: int = 64,
                 device: int = 0):
        if max_suffix_length is None or int(max_suffix_length) < 1:
            raise ValueError("max_suffix_length must be >= 1")
        self.max_suffix_length = int(max_suffix_length)
        self.device = int(device)
        self._index = None
        self._mode = None
        self.columns = None
        if documents is not None and csv_file is not None:
            raise ValueError("pass either documents= or csv_file=, not both")
        if documents is not None:
            self.construct_truncated_suffix_array_documents(documents)
        elif csv_file is not None:
            if search_column is None:
                raise ValueError("search_column is required with csv_file")
            self.construct_truncated_suffix_array_from_csv(csv_file, search_column)

    # -- construction ---------------------------------------------------------------------------
    def construct_truncated_suffix_array_documents(self, documents):
        """pyx:129-180: text = '\\n'.join(documents), lower-cased, one truncated SA over it."""
        if not isinstance(documents, list):
            try:
                documents = list(documents)
            except Exception:
                raise ValueError("Documents must be a list of strings")
        self._documents = documents
        encoded = [d.encode("utf-8") for d in documents]
        text = _ascii_lower(b"\n".join(encoded))
        lens = np.fromiter((len(e) for e in encoded), dtype=np.int64, count=len(encoded))
        # start offset of every document in the joined text
        self._row_starts = np.concatenate([[0], np.cumsum(lens + 1)[:-1]]).astype(np.int64) if len(encoded) else np.zeros(0, np.int64)
        self._set_text(text)
        self._mode = "documents"

    def construct_truncated_suffix_array_from_csv(self, filename: str, search_column: str):
        """pyx:183-207 / engine.c:461-654: index one column of a CSV file (RFC-4180 quoting)."""
        from .csv_ingest import extract_column
        import threading
        self.csv_filename = filename
        # the HIP runtime of this process (context, code object: ~0.2 s the first time) comes up while the host
        # parses the file; the extractor runs outside the GIL
        timing = os.environ.get("SA_HIP_CSV_TIMING", "0") not in ("", "0")
        t0 = time.perf_counter()
        warm = threading.Thread(target=_warm_device, args=(self.device,), daemon=True)
        warm.start()
        try:
            col = extract_column(filename, search_column)
            t1 = time.perf_counter()
        finally:
            warm.join()
        t2 = time.perf_counter()
        self.columns = col.columns
        self._row_starts = col.text_row_starts      # offset of every row's field in the column text
        self._row_file_offsets = col.row_file_offsets
        self._set_text(col.text_array)              # a view of the extractor's own buffer: no copy
        self._mode = "csv"
        if timing:
            print("[sa_hip csv] extract %.3f s, wait for the device %.3f s, create + upload + build %.3f s" % (
                t1 - t0, t2 - t1, time.perf_counter() - t2), file=sys.stderr)

    def _set_text(self, text):
        """text: bytes or a uint8 array"""
        if len(text) > 0xFFFFFFFE:
            raise ValueError("text exceeds 2^32 - 2 bytes (one index per device)")
        self._text_len = len(text)
        self._text_bytes = text
        if self._index is not None:
            self._index.close()
        self._index = _capi.DeviceIndex(max(len(text), 1), self.device)
        self._index.build(text, self.max_suffix_length)

    # -- query ------------------------------------------------------------------------------------
    def query_ranges(self, substrings):
        """Batched get_substring_positions: (first, second) per pattern, reference conventions."""
        pats = [_ascii_lower(s.encode("utf-8")) if isinstance(s, str) else _ascii_lower(bytes(s)) for s in substrings]
        return self._index.query_batch(pats)

    def _rows_for_range(self, first, second, k, first_hits=None):
        if first == _capi.UINT32_MAX or ((second - first + 1) & 0xFFFFFFFF) == 0:
            return np.zeros(0, np.int64)
        count = int(second) - int(first) + 1
        rows = []
        seen = set()
        # hits arrive in SA order; walk them in slabs until k distinct rows are found
        # (first_hits: SA[first .. first + len) already fetched by query_hits)
        pos = int(first)
        end = int(first) + count
        slab = max(4 * k, 1024)
        while pos < end and len(rows) < k:
            take = min(slab, end - pos)
            if first_hits is not None and pos == int(first) and len(first_hits):
                take = min(take, len(first_hits))
                hits = first_hits[:take].astype(np.int64)
            else:
                hits = self._index.sa_range(pos, take).astype(np.int64)
            ids = np.searchsorted(self._row_starts, hits, side="right") - 1
            # distinct rows in order of first appearance (vectorised; the Python set only spans slabs)
            _, first_at = np.unique(ids, return_index=True)
            for r in ids[np.sort(first_at)].tolist():
                if r not in seen:
                    seen.add(r)
                    rows.append(r)
                    if len(rows) == k:
                        break
            pos += take
        return np.asarray(rows, dtype=np.int64)

    def _materialise(self, rows):
        if self._mode == "documents":
            return [self._documents[r] for r in rows.tolist()]
        # the file is mapped once (the reference re-opens it and does one fseek + fread per row,
        # engine.c:1334-1390); a row without a quote character is split directly, the rest goes through the csv module
        mm = self._csv_map()
        off = self._row_file_offsets
        cols = self.columns
        out = []
        for r in rows.tolist():
            raw = mm[int(off[r]):int(off[r + 1])]
            if b'"' in raw:
                rec = next(_csv.reader(io.StringIO(raw.decode("utf-8", "replace"))))
            else:
                rec = raw.decode("utf-8", "replace").rstrip("\r\n").split(",")
            out.append(dict(zip(cols, rec)))
        return out

    def _csv_map(self):
        mm = getattr(self, "_csv_mm", None)
        if mm is None:
            import mmap
            self._csv_fh = open(self.csv_filename, "rb")
            mm = self._csv_mm = mmap.mmap(self._csv_fh.fileno(), 0, access=mmap.ACCESS_READ)
        return mm

    def query_records(self, substring: str, k: int = 1000):
        """pyx:209-267: records containing `substring` (case-insensitive ASCII), at most k."""
        if substring == "":
            return []
        if self._index is None:
            raise RuntimeError("index not built")
        # one call fetches the range and the first hits (sa_hip_index_query_hits: no copy calls, one synchronisation)
        pat = _ascii_lower(substring.encode("utf-8")) if isinstance(substring, str) else _ascii_lower(bytes(substring))
        (first, second), hits = self._index.query_hits(pat, min(max(4 * k, 1024), 4096))
        return self._materialise(self._rows_for_range(int(first), int(second), k, hits))

    def query_records_batch(self, substrings, k: int = 1000):
        if self._index is None:
            raise RuntimeError("index not built")
        res = [None] * len(substrings)
        live = [i for i, s in enumerate(substrings) if s != ""]
        ranges = self.query_ranges([substrings[i] for i in live]) if live else []
        for i in range(len(substrings)):
            if substrings[i] == "":
                res[i] = []
        for j, i in enumerate(live):
            rows = self._rows_for_range(int(ranges[j]["first"]), int(ranges[j]["second"]), k)
            res[i] = self._materialise(rows)
        return res

    # -- persistence (the reference's save/load is half-built: engine.c:1098-1165,
    #    commented-out pyx:310-423).  Versioned directory: meta.json + raw little-endian arrays. ------
    FORMAT_VERSION = 1

    def save(self, directory: str):
        """Write the index (text, uint32 suffix array, row tables) so that load() needs no rebuild."""
        import json
        if self._index is None:
            raise RuntimeError("index not built")
        os.makedirs(directory, exist_ok=True)
        n = self._index.n
        self._index.sa_u32().tofile(os.path.join(directory, "sa.u32"))
        # the indexed (lower-cased) text is re-derived from the source for documents, stored for CSV
        meta = {"format": "suffixarray_amd", "version": self.FORMAT_VERSION, "mode": self._mode, "n": n,
                "max_suffix_length": self.max_suffix_length, "columns": self.columns}
        np.asarray(self._row_starts, dtype=np.int64).tofile(os.path.join(directory, "row_starts.i64"))
        with open(os.path.join(directory, "text.u8"), "wb") as f:
            f.write(memoryview(self._text_bytes))
        if self._mode == "documents":
            with open(os.path.join(directory, "documents.json"), "w") as f:
                json.dump(self._documents, f)
        else:
            meta["csv_filename"] = os.path.abspath(self.csv_filename)
            np.asarray(self._row_file_offsets, dtype=np.int64).tofile(os.path.join(directory, "row_file_offsets.i64"))
        with open(os.path.join(directory, "meta.json"), "w") as f:
            json.dump(meta, f)

    @classmethod
    def load(cls, directory: str, device: int = 0):
        """Re-open a saved index: uploads text + SA (sa_hip_index_load), no construction."""
        import json
        with open(os.path.join(directory, "meta.json")) as f:
            meta = json.load(f)
        if meta.get("format") != "suffixarray_amd" or meta.get("version") != cls.FORMAT_VERSION:
            raise ValueError("not a suffixarray_amd index of a supported version")
        self = cls(max_suffix_length=meta["max_suffix_length"], device=device)
        self._mode = meta["mode"]
        self.columns = meta["columns"]
        text = np.fromfile(os.path.join(directory, "text.u8"), dtype=np.uint8)
        sa = np.fromfile(os.path.join(directory, "sa.u32"), dtype=np.uint32)
        if text.size != meta["n"] or sa.size != meta["n"]:
            raise ValueError("index files are truncated")
        self._row_starts = np.fromfile(os.path.join(directory, "row_starts.i64"), dtype=np.int64)
        self._text_bytes = text.tobytes()
        self._text_len = text.size
        if self._mode == "documents":
            with open(os.path.join(directory, "documents.json")) as f:
                self._documents = json.load(f)
        else:
            self.csv_filename = meta["csv_filename"]
            self._row_file_offsets = np.fromfile(os.path.join(directory, "row_file_offsets.i64"), dtype=np.int64)
        self._index = _capi.DeviceIndex(max(text.size, 1), self.device)
        self._index.load(text, sa, self.max_suffix_length)
        return self

    def close(self):
        if self._index is not None:
            self._index.close()
            self._index = None
        mm = getattr(self, "_csv_mm", None)
        if mm is not None:
            mm.close()
            self._csv_fh.close()
            self._csv_mm = None
