// records.hpp -- record retrieval at the C seam (host side): hits -> the rows that contain them ->
// malloc'ed row strings, with the reference's calling conventions.
//
// Replaces get_matching_records_file (engine.c:1326-1390) + rfc4180_seek_backward/forward_newline (engine.c:1217-1323)
// and get_matching_records (engine.c:1168-1215), bound by the reference's Cython layer at pyx:87-101 and called per
// query at pyx:224-232.  The reference finds a hit's row by seeking backwards and forwards from the hit's FILE offset
// to the next unquoted newline (4 KiB page reads, one fopen per call); here the extractor has already recorded where
// every row starts (csv_ingest.hpp), so a hit's row is one binary search over the row table and a row is one memcpy
// out of the memory-mapped file.  Decisions: a row that contains the pattern several times is returned
// ONCE (the reference returns it once per hit), rows come back whole (the reference drops the last character,
// engine.c:1314, and beyond the file's first page also the first one), a miss returns nothing (the reference runs
// into undefined behaviour, engine.c:1347-1356).
#pragma once
#include <algorithm>
#include <string>
#include <unordered_set>
#include <vector>

#include "common.hpp"
#include "csv_ingest.hpp"

namespace sa {

// distinct rows of the hits SA[first .. second], in order of first appearance, at most k.
// fetch(pos, count, out) copies SA[pos .. pos + count) to the host; first_hits = the hits already fetched with the range.
template <class Fetch>
inline int distinct_rows(const HostU64Array& row_starts, sa_hip_pair_u32 range, u32 k, const u32* first_hits, u32 n_first,
                         Fetch&& fetch, std::vector<u64>& rows) {
    rows.clear();
    if (k == 0 || row_starts.empty() || range.first == 0xFFFFFFFFu || (u32)(range.second - range.first + 1u) == 0u) return 0;
    const u64 end = (u64)range.second + 1;
    u64 pos = range.first;
    std::unordered_set<u64> seen;
    std::vector<u32> slab;
    const u64 slab_len = std::max<u64>(4ull * k, 1024);
    while (pos < end && rows.size() < k) {
        u64 take = std::min<u64>(slab_len, end - pos);
        const u32* hits;
        if (pos == range.first && n_first) { take = std::min<u64>(take, n_first); hits = first_hits; }
        else {
            slab.resize((size_t)take);
            int rc = fetch(pos, take, slab.data());
            if (rc) return rc;
            hits = slab.data();
        }
        for (u64 i = 0; i < take && rows.size() < k; ++i) {
            const u64 p = hits[i];
            const u64 r = (u64)(std::upper_bound(row_starts.begin(), row_starts.end(), p) - row_starts.begin()) - 1;   // row_starts[0] = 0
            if (seen.insert(r).second) rows.push_back(r);
        }
        pos += take;
    }
    return 0;
}

// row [b, e) of a mapped CSV file without its line terminator, as a malloc'ed NUL-terminated string (engine.c:1382:
// the callee allocates every record, the caller frees them, pyx:262-265)
inline char* dup_row(const u8* base, u64 b, u64 e) {
    while (e > b && (base[e - 1] == '\n' || base[e - 1] == '\r')) --e;
    char* s = static_cast<char*>(malloc((size_t)(e - b) + 1));
    if (!s) return nullptr;
    memcpy(s, base + b, (size_t)(e - b));
    s[e - b] = '\0';
    return s;
}

}  // namespace sa
