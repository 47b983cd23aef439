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

#include <sched.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
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

// One worker's share of the file: the records that START inside [lo, hi).
struct CsvPart {
    std::vector<u8> text;
    std::vector<u64> starts;   // relative to this part's text
    std::vector<u64> offs;     // file offsets of the rows
    u64 last_end = 0;
};

// Fast path of csv_parse_record for one wanted field: the field's bytes go straight into `text`
// (lower-cased, embedded newlines as spaces).  Same state machine, same return values, but it moves in RUNS: outside
// quotes up to the next of , " \n \r (table look-up), inside quotes up to the next " (memchr); a run of the wanted
// field is appended with one resize and a branch-free lower-casing loop.
struct CsvSpecial {
    u8 t[256];
    constexpr CsvSpecial() : t() { t[(u8)','] = 1; t[(u8)'"'] = 1; t[(u8)'\n'] = 1; t[(u8)'\r'] = 1; }
};
inline void csv_append_lower(std::vector<u8>& text, const u8* src, u64 len, bool newline_to_space) {
    const size_t o = text.size();
    text.resize(o + len);
    u8* dst = text.data() + o;
    for (u64 k = 0; k < len; ++k) {
        const u8 c = src[k];
        dst[k] = (u8)(c + (((u8)(c - 65u) < 26u) ? 32u : 0u));
    }
    if (newline_to_space)
        for (u64 k = 0; k < len; ++k) if (dst[k] == '\n') dst[k] = ' ';
}
inline u32 csv_parse_record_into(CsvRecordCursor& cur, u32 want, std::vector<u8>& text, u64* row_start, u64* row_end,
                                 bool* first_field_empty) {
    static constexpr CsvSpecial special{};
    const u8* d = cur.d;
    const u64 n = cur.n;
    u64 i = cur.i;
    *row_start = i;
    u32 fidx = 0;
    bool in_quotes = false, f0_empty = true;
    while (true) {
        if (i >= n) { ++fidx; break; }
        if (in_quotes) {
            const u8* q = static_cast<const u8*>(memchr(d + i, '"', n - i));
            const u64 j = q ? (u64)(q - d) : n;
            if (j > i) {
                if (fidx == want) csv_append_lower(text, d + i, j - i, true);
                if (fidx == 0) f0_empty = false;
                i = j;
            }
            if (i >= n) continue;   // unterminated quote: the record ends with the file
            if (i + 1 < n && d[i + 1] == '"') {   // doubled quote = one literal quote
                if (fidx == want) text.push_back('"');
                if (fidx == 0) f0_empty = false;
                i += 2;
                continue;
            }
            in_quotes = false; ++i; continue;
        }
        u64 j = i;
        while (j < n && !special.t[d[j]]) ++j;
        if (j > i) {
            if (fidx == want) csv_append_lower(text, d + i, j - i, false);
            if (fidx == 0) f0_empty = false;
            i = j;
            if (i >= n) continue;
        }
        const u8 c = d[i];
        if (c == '"') { in_quotes = true; ++i; }
        else if (c == ',') { ++fidx; ++i; }
        else {   // '\n' or '\r'
            ++fidx;
            if (c == '\r' && i + 1 < n && d[i + 1] == '\n') ++i;
            ++i;
            break;
        }
    }
    cur.i = i;
    *row_end = i;
    *first_field_empty = f0_empty;
    return fidx;
}

// Large, freshly allocated buffers that are about to be written once: ask for transparent huge pages (the boxes run THP in
// `madvise` mode) -- a 4 KB first-touch fault per page was a measurable part of parse, merge and of giving the memory back.
inline void csv_advise_huge(const void* p, size_t bytes) {
    const uintptr_t a = ((uintptr_t)p + 4095u) & ~(uintptr_t)4095u, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095u;
    if (e > a && e - a >= ((size_t)4 << 20)) (void)madvise(reinterpret_cast<void*>(a), e - a, MADV_HUGEPAGE);
}

inline void csv_parse_range(const u8* d, u64 n, u64 begin, u64 hi, u32 ci, CsvPart& part) {
    CsvRecordCursor cur{d, n, begin};
    u64 rs, re;
    bool f0e;
    part.last_end = begin;
    part.text.reserve((size_t)((hi - begin) / 2 + 64));
    part.starts.reserve((size_t)((hi - begin) / 24 + 16));
    part.offs.reserve((size_t)((hi - begin) / 24 + 16));
    csv_advise_huge(part.text.data(), part.text.capacity());
    csv_advise_huge(part.starts.data(), part.starts.capacity() * 8);
    csv_advise_huge(part.offs.data(), part.offs.capacity() * 8);
    while (cur.i < hi) {
        // Fast path: a row without a quote character (and without a bare '\r') needs no state machine -- it ends at
        // the next '\n', its fields are separated by every ','.  Three memchr sweeps over the ~30 bytes of the row
        // + one per comma up to the wanted field, against ~250 ns per row in the run-based parser below.
        {
            const u64 i = cur.i;
            const u8* p = d + i;
            const u8* nl = static_cast<const u8*>(memchr(p, '\n', n - i));
            const u64 len = nl ? (u64)(nl - p) : n - i;
            if (!memchr(p, '"', len)) {
                const u8* cr = static_cast<const u8*>(memchr(p, '\r', len));
                if (!cr || (u64)(cr - p) == len - 1) {   // no '\r', or the one of a CRLF / a last '\r' at the end of the file
                    const u64 clen = cr ? len - 1 : len;
                    const u64 next = nl ? i + len + 1 : n;
                    cur.i = next;
                    if (clen == 0) continue;   // blank line
                    const u8* f = p;
                    const u8* fend = p + clen;
                    u32 k = 0;
                    for (; k < ci; ++k) {
                        const u8* c = static_cast<const u8*>(memchr(f, ',', (size_t)(fend - f)));
                        if (!c) break;
                        f = c + 1;
                    }
                    const u64 tpos = part.text.size();
                    if (k == ci) {
                        const u8* c = static_cast<const u8*>(memchr(f, ',', (size_t)(fend - f)));
                        const u8* g = c ? c : fend;
                        const u64 flen = (u64)(g - f);
                        part.text.resize(tpos + flen + 1);
                        u8* dst = part.text.data() + tpos;
                        for (u64 q = 0; q < flen; ++q) { const u8 ch = f[q]; dst[q] = (u8)(ch + (((u8)(ch - 65u) < 26u) ? 32u : 0u)); }
                        dst[flen] = '\n';
                    } else part.text.push_back('\n');
                    part.starts.push_back(tpos);
                    part.offs.push_back(i);
                    part.last_end = next;
                    continue;
                }
            }
        }
        const u64 tpos = part.text.size();
        const u32 nf = csv_parse_record_into(cur, ci, part.text, &rs, &re, &f0e);
        if (nf == 1 && f0e) continue;  // blank line (nothing was appended)
        part.starts.push_back(tpos);
        part.offs.push_back(rs);
        part.text.push_back('\n');
        part.last_end = re;
    }
}

// fn(k) for k in [0, parts) on one thread each.  A thread that cannot be created (process / cgroup limit:
// std::system_error) has its share run on the calling thread instead; every thread that was started is joined
// before this returns or unwinds, so an exception of fn's (bad_alloc) reaches the extern "C" wrapper's catch
// instead of std::terminate.
template <class F>
inline void csv_parallel(u64 parts, F&& fn) {
    struct Group {
        std::vector<std::thread> th;
        ~Group() { for (auto& t : th) if (t.joinable()) t.join(); }
    } g;
    std::exception_ptr first;
    std::mutex mu;
    auto guarded = [&](u64 k) {
        try { fn(k); }
        catch (...) { std::lock_guard<std::mutex> l(mu); if (!first) first = std::current_exception(); }
    };
    g.th.reserve((size_t)parts);
    for (u64 k = 0; k < parts; ++k) {
        try { g.th.emplace_back(guarded, k); }
        catch (const std::system_error&) { guarded(k); }
    }
    for (auto& t : g.th) t.join();
    if (first) std::rethrow_exception(first);
}

// Multi-threaded: the file is cut into byte ranges; the quote parity in front of every range (a '"'
// toggles it; the doubled quote of RFC-4180 toggles twice) tells whether a newline there is a row
// terminator, so every worker can find the first row that starts inside its range on its own.
inline int csv_extract_column(const char* path, const char* column, sa_hip_csv_column* out) {
    memset(out, 0, sizeof *out);
    // SA_HIP_CSV_TIMING=1: phase times on stderr
    const bool timing = diag_env("SA_HIP_CSV_TIMING") && atoi(diag_env("SA_HIP_CSV_TIMING")) != 0;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sa_hip csv] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: cannot open file", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: fstat failed", path); }
    const u64 n = (u64)sb.st_size;
    if (n == 0) { close(fd); return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: empty CSV file"); }
    void* map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(SA_HIP_ENOMEM, "sa_hip_csv_extract_column: mmap failed", path);
    struct Unmap {   // also on the unwinding path (an exception of a worker ends in the extern "C" wrapper's catch)
        void* p; u64 n;
        void release() { if (p) munmap(p, n); p = nullptr; }
        ~Unmap() { release(); }
    } mapping{map, n};
    const u8* d = static_cast<const u8*>(map);
    CsvRecordCursor cur{d, n, 0};

    std::vector<std::string> header;
    u64 rs, re;
    bool f0e;
    csv_parse_record<true>(cur, &header, 0, nullptr, &rs, &re, &f0e);
    int ci = -1;
    for (size_t k = 0; k < header.size(); ++k) if (header[k] == column) { ci = (int)k; break; }
    if (ci < 0) { mapping.release(); return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: column not found", column); }
    const u64 body = re;   // first byte after the header row

    // ranges
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("SA_HIP_CSV_THREADS")) hw = (unsigned)atoi(e);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const unsigned a = (unsigned)CPU_COUNT(&set); if (a && a < hw) hw = a; }
    if (hw < 1) hw = 1;
    if (hw > 64) hw = 64;
    u64 parts_n = (n - body) / (4u << 20) + 1;   // >= 4 MiB per worker
    if (parts_n > hw) parts_n = hw;
    std::vector<u64> lo(parts_n + 1);
    for (u64 k = 0; k <= parts_n; ++k) lo[k] = body + (n - body) * k / parts_n;
    // pass 1: quote counts per range -> parity in front of each range
    std::vector<u64> quotes(parts_n, 0);
    {
        csv_parallel(parts_n, [&](u64 k) { u64 c = 0; for (u64 i = lo[k]; i < lo[k + 1]; ++i) c += (d[i] == '"'); quotes[k] = c; });
    }
    lap("quote parity (first touch)");
    // start of the first record of every range: after the first newline outside quotes at or after lo[k]
    // (a record that starts exactly at lo[k] is recognised by the terminator just before it)
    std::vector<u64> begin(parts_n + 1, n);
    begin[0] = body;
    {
        u64 parity = 0;
        for (u64 k = 1; k < parts_n; ++k) {
            parity ^= (quotes[k - 1] & 1);
            u64 i = lo[k];
            bool inq = parity != 0;
            // step back one byte: if the previous byte terminates a row (outside quotes), lo[k] starts a row
            u64 b = n;
            if (!inq && i > body && (d[i - 1] == '\n' || (d[i - 1] == '\r' && !(i < n && d[i] == '\n')))) b = i;
            for (; b == n && i < n; ++i) {
                const u8 c = d[i];
                if (c == '"') inq = !inq;
                else if (!inq && c == '\n') b = i + 1;
                else if (!inq && c == '\r' && !(i + 1 < n && d[i + 1] == '\n')) b = i + 1;
            }
            begin[k] = b;
        }
        begin[parts_n] = n;
        for (u64 k = parts_n; k-- > 1;) if (begin[k] > begin[k + 1]) begin[k] = begin[k + 1];
    }
    // pass 2: parse the rows that start in [begin[k], begin[k+1])
    std::vector<CsvPart> part(parts_n);
    {
        csv_parallel(parts_n, [&](u64 k) { if (begin[k] < begin[k + 1]) csv_parse_range(d, n, begin[k], begin[k + 1], (u32)ci, part[k]); });
    }
    lap("parse");
    mapping.release();

    u64 tlen = 0, rows = 0, last_end = 0;
    for (auto& p : part) { tlen += p.text.size(); rows += p.starts.size(); if (!p.starts.empty()) last_end = p.last_end; }
    out->text_len = tlen;
    out->num_rows = rows;
    out->num_columns = (u32)header.size();
    out->column_index = (u32)ci;
    size_t names_len = 0;
    for (auto& h : header) names_len += h.size() + 1;
    out->text = (uint8_t*)malloc(tlen ? tlen : 1);
    out->row_text_starts = (uint64_t*)malloc((rows ? rows : 1) * sizeof(uint64_t));
    out->row_file_offsets = (uint64_t*)malloc((rows + 1) * sizeof(uint64_t));
    out->column_names = (char*)malloc(names_len ? names_len : 1);
    if (!out->text || !out->row_text_starts || !out->row_file_offsets || !out->column_names) {
        free(out->text); free(out->row_text_starts); free(out->row_file_offsets); free(out->column_names);
        memset(out, 0, sizeof *out);
        return fail(SA_HIP_ENOMEM, "sa_hip_csv_extract_column: out of host memory");
    }
    csv_advise_huge(out->text, tlen);
    csv_advise_huge(out->row_text_starts, rows * 8);
    csv_advise_huge(out->row_file_offsets, (rows + 1) * 8);
    // every worker copies its own part into place (the output arrays are first touched in parallel too)
    {
        std::vector<u64> toff(parts_n + 1, 0), roff(parts_n + 1, 0);
        for (u64 k = 0; k < parts_n; ++k) { toff[k + 1] = toff[k] + part[k].text.size(); roff[k + 1] = roff[k] + part[k].starts.size(); }
        csv_parallel(parts_n, [&](u64 k) {
                CsvPart& p = part[k];
                if (!p.text.empty()) memcpy(out->text + toff[k], p.text.data(), p.text.size());
                for (size_t r = 0; r < p.starts.size(); ++r) {
                    out->row_text_starts[roff[k] + r] = toff[k] + p.starts[r];
                    out->row_file_offsets[roff[k] + r] = p.offs[r];
                }
                std::vector<u8>().swap(p.text);   // release the part as soon as it is copied
                std::vector<u64>().swap(p.starts);
                std::vector<u64>().swap(p.offs);
            });
    }
    lap("merge into output arrays");
    out->row_file_offsets[rows] = rows ? last_end : 0;
    char* pn = out->column_names;
    for (auto& h : header) { memcpy(pn, h.c_str(), h.size() + 1); pn += h.size() + 1; }
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
