"""CSV column extractor (host side; SURVEY.md 8(f)-1, reference engine.c:26-96, 461-654).

Produces what the device build needs: the search column lower-cased with a '\\n' after every
field, plus per-row offsets.  RFC-4180: fields may be quoted, quotes are doubled inside quoted
fields, quoted fields may contain commas and newlines.  Differences from the reference, by
decision: the header row is NOT indexed (engine.c indexes it like a record), file
offsets are 64-bit (the reference truncates to uint32).
"""
import numpy as np

from suffixarray_amd.index import _ascii_lower


class ColumnText:
    """columns, text (bytes), text_row_starts, row_file_offsets.  text_array is the same text as a uint8 array; the
    native extractor fills only that (a view of its own buffer) and `text` is made from it on first use."""
    __slots__ = ("columns", "_text", "text_array", "text_row_starts", "row_file_offsets")

    def __init__(self):
        self._text = None
        self.text_array = None

    @property
    def text(self):
        if self._text is None and self.text_array is not None:
            self._text = self.text_array.tobytes()
        return self._text

    @text.setter
    def text(self, value):
        self._text = value
        self.text_array = np.frombuffer(value, dtype=np.uint8)


def _parse_rows(data: bytes):
    """Yield (row_start, row_end_exclusive_incl_newline, fields as list[bytes]) per record."""
    n = len(data)
    i = 0
    while i < n:
        row_start = i
        fields = []
        cur = bytearray()
        in_quotes = False
        while True:
            if i >= n:
                fields.append(bytes(cur))
                break
            c = data[i]
            if in_quotes:
                if c == 0x22:  # '"'
                    if i + 1 < n and data[i + 1] == 0x22:
                        cur.append(0x22)
                        i += 2
                        continue
                    in_quotes = False
                    i += 1
                    continue
                cur.append(c)
                i += 1
                continue
            if c == 0x22:
                in_quotes = True
                i += 1
            elif c == 0x2C:  # ','
                fields.append(bytes(cur))
                cur = bytearray()
                i += 1
            elif c == 0x0A or c == 0x0D:
                fields.append(bytes(cur))
                if c == 0x0D and i + 1 < n and data[i + 1] == 0x0A:
                    i += 1
                i += 1
                break
            else:
                cur.append(c)
                i += 1
        yield row_start, i, fields


def extract_column(filename: str, search_column: str) -> ColumnText:
    """Native extractor (libsa_hip.so, csrc/csv_ingest.hpp); extract_column_py is the same state
    machine in Python and serves as its reference in the tests."""
    from suffixarray_amd import _capi
    try:
        names, text, starts, offs = _capi.csv_extract_column(filename, search_column, copy=False)
    except _capi.SaHipError as e:
        raise ValueError(str(e))
    out = ColumnText()
    out.columns, out.text_array, out.text_row_starts, out.row_file_offsets = names, text, starts, offs
    return out


def extract_column_py(filename: str, search_column: str) -> ColumnText:
    with open(filename, "rb") as f:
        data = f.read()
    rows = _parse_rows(data)
    try:
        _, _, header = next(rows)
    except StopIteration:
        raise ValueError("empty CSV file")
    columns = [h.decode("utf-8") for h in header]
    if search_column not in columns:
        raise ValueError(f"Column {search_column} not found in CSV file")
    ci = columns.index(search_column)
    parts = []
    starts = []
    offs = []
    pos = 0
    end = 0
    for rs, re_, fields in rows:
        if len(fields) == 1 and fields[0] == b"":
            continue  # blank line
        field = fields[ci] if ci < len(fields) else b""
        field = _ascii_lower(field).replace(b"\n", b" ")  # keep '\n' as the row terminator only
        starts.append(pos)
        offs.append(rs)
        parts.append(field)
        pos += len(field) + 1
        end = re_
    out = ColumnText()
    out.columns = columns
    out.text = b"\n".join(parts) + (b"\n" if parts else b"")
    out.text_row_starts = np.asarray(starts, dtype=np.int64)
    offs.append(end)
    out.row_file_offsets = np.asarray(offs, dtype=np.int64)
    return out
