// csv_ingest.hpp -- host-side CSV column extractor (reference engine.c:26-96 header
// parser, engine.c:461-654 mmap builder).  Produces what the device build needs: the search column
// lower-cased with a '\n' after every field, the text offset of every row's field and the file offset
// of every row.  RFC-4180: quoted fields, doubled quotes, commas / newlines inside quotes.
// Decisions: the header row is NOT indexed (engine.c indexes it like a record), file
// offsets are 64-bit (the reference truncates to uint32), a newline inside a quoted field becomes a
// space in the index text (so '\n' stays the row terminator), blank lines are skipped.
// suffixarray_amd/csv_ingest.py holds the same state machine in Python; tests compare the two.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

namespace sa {

struct CsvRecordCursor {
    const u8* d;
    u64 n;
    u64 i = 0;
};

// Parses one record starting at cur.i.  Fields are appended to `fields` when capture_all, otherwise
// only field `want` is appended to `one`.  Returns the number of fields; sets row_start / row_end.
template <bool CAPTURE_ALL>
inline u32 csv_parse_record(CsvRecordCursor& cur, std::vector<std::string>* fields, u32 want, std::string* one,
                            u64* row_start, u64* row_end, bool* first_field_empty) {
    const u8* d = cur.d;
    const u64 n = cur.n;
    u64 i = cur.i;
    *row_start = i;
    u32 fidx = 0;
    bool in_quotes = false;
    std::string curf;
    bool f0_empty = true;
    auto put = [&](u8 c) {
        if (CAPTURE_ALL) curf.push_back((char)c);
        else if (fidx == want) one->push_back((char)c);
        if (fidx == 0) f0_empty = false;
    };
    auto end_field = [&]() {
        if (CAPTURE_ALL) { fields->push_back(curf); curf.clear(); }
        ++fidx;
    };
    while (true) {
        if (i >= n) { end_field(); break; }
        const u8 c = d[i];
        if (in_quotes) {
            if (c == '"') {
                if (i + 1 < n && d[i + 1] == '"') { put('"'); i += 2; continue; }
                in_quotes = false; ++i; continue;
            }
            put(c); ++i; continue;
        }
        if (c == '"') { in_quotes = true; ++i; }
        else if (c == ',') { end_field(); ++i; }
        else if (c == '\n' || c == '\r') {
            end_field();
            if (c == '\r' && i + 1 < n && d[i + 1] == '\n') ++i;
            ++i;
            break;
        } else { put(c); ++i; }
    }
    cur.i = i;
    *row_end = i;
    *first_field_empty = f0_empty;
    return fidx;
}

inline int csv_extract_column(const char* path, const char* column, sa_hip_csv_column* out) {
    memset(out, 0, sizeof *out);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: cannot open file", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: fstat failed", path); }
    const u64 n = (u64)sb.st_size;
    if (n == 0) { close(fd); return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: empty CSV file"); }
    void* map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(SA_HIP_ENOMEM, "sa_hip_csv_extract_column: mmap failed", path);
    (void)madvise(map, n, MADV_SEQUENTIAL);
    CsvRecordCursor cur{static_cast<const u8*>(map), n, 0};

    std::vector<std::string> header;
    u64 rs, re;
    bool f0e;
    csv_parse_record<true>(cur, &header, 0, nullptr, &rs, &re, &f0e);
    int ci = -1;
    for (size_t k = 0; k < header.size(); ++k) if (header[k] == column) { ci = (int)k; break; }
    if (ci < 0) { munmap(map, n); return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: column not found", column); }

    std::vector<u8> text;
    text.reserve((size_t)(n / 2));
    std::vector<u64> starts, offs;
    std::string field;
    u64 last_end = re;
    while (cur.i < n) {
        field.clear();
        const u32 nf = csv_parse_record<false>(cur, nullptr, (u32)ci, &field, &rs, &re, &f0e);
        if (nf == 1 && f0e) continue;  // blank line
        starts.push_back((u64)text.size());
        offs.push_back(rs);
        for (char ch : field) {
            u8 c = (u8)ch;
            if (c >= 65 && c <= 90) c += 32;   // suffix_array.pyx:103-107: ASCII only
            if (c == '\n') c = ' ';
            text.push_back(c);
        }
        text.push_back('\n');
        last_end = re;
    }
    munmap(map, n);
    offs.push_back(starts.empty() ? 0 : last_end);

    out->text_len = text.size();
    out->num_rows = starts.size();
    out->num_columns = (u32)header.size();
    out->column_index = (u32)ci;
    size_t names_len = 0;
    for (auto& h : header) names_len += h.size() + 1;
    out->text = (uint8_t*)malloc(text.size() ? text.size() : 1);
    out->row_text_starts = (uint64_t*)malloc((starts.size() ? starts.size() : 1) * sizeof(uint64_t));
    out->row_file_offsets = (uint64_t*)malloc(offs.size() * sizeof(uint64_t));
    out->column_names = (char*)malloc(names_len ? names_len : 1);
    if (!out->text || !out->row_text_starts || !out->row_file_offsets || !out->column_names) {
        free(out->text); free(out->row_text_starts); free(out->row_file_offsets); free(out->column_names);
        memset(out, 0, sizeof *out);
        return fail(SA_HIP_ENOMEM, "sa_hip_csv_extract_column: out of host memory");
    }
    if (!text.empty()) memcpy(out->text, text.data(), text.size());
    if (!starts.empty()) memcpy(out->row_text_starts, starts.data(), starts.size() * sizeof(u64));
    memcpy(out->row_file_offsets, offs.data(), offs.size() * sizeof(u64));
    char* p = out->column_names;
    for (auto& h : header) { memcpy(p, h.c_str(), h.size() + 1); p += h.size() + 1; }
    return 0;
}

inline void csv_free(sa_hip_csv_column* c) {
    if (!c) return;
    free(c->text); free(c->row_text_starts); free(c->row_file_offsets); free(c->column_names);
    memset(c, 0, sizeof *c);
}

// Synthetic CSV of BASELINE config 5: rows `id,company_name,country`;
// company_name = 1..3 words of a 50 000-word synthetic vocabulary (skewed towards low ranks) +
// optional suffix; `, Inc.` forces RFC-4180 quoting; some names are capitalised.
inline int synth_csv(const char* path, u64 rows, u64 seed) {
    FILE* f = fopen(path, "wb");
    if (!f) return fail(SA_HIP_EINVAL, "sa_hip_synth_csv: cannot create file", path);
    u64 s = seed ? seed : 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const int V = 50000;
    std::vector<std::string> vocab(V);
    for (int i = 0; i < V; ++i) {
        const int len = 2 + (int)(rnd() % 11);
        std::string w(len, 'a');
        for (int k = 0; k < len; ++k) w[k] = (char)('a' + rnd() % 26);
        vocab[i] = w;
    }
    static const char* suffix[] = {"", "", "", " inc", " llc", " ltd", ", Inc."};
    static const char* country[] = {"US", "DE", "GB", "FR", "JP", "IN", "BR", "CA"};
    std::string buf;
    buf.reserve(1 << 22);
    buf += "id,company_name,country\n";
    for (u64 r = 0; r < rows; ++r) {
        std::string name;
        const int nw = 1 + (int)(rnd() % 3);
        for (int k = 0; k < nw; ++k) {
            const double u = (double)(rnd() >> 11) / 9007199254740992.0;
            const int idx = (int)(V * u * u * u);
            std::string w = vocab[idx < V ? idx : V - 1];
            if ((rnd() & 3) == 0) w[0] = (char)(w[0] - 32);
            if (k) name += ' ';
            name += w;
        }
        const char* sx = suffix[rnd() % 7];
        name += sx;
        buf += std::to_string(r + 1);
        buf += ',';
        if (name.find(',') != std::string::npos) { buf += '"'; buf += name; buf += '"'; }
        else buf += name;
        buf += ',';
        buf += country[rnd() % 8];
        buf += '\n';
        if (buf.size() > (1u << 22) - 256) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
    }
    fwrite(buf.data(), 1, buf.size(), f);
    fclose(f);
    return 0;
}

}  // namespace sa
