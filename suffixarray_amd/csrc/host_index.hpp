// host_index.hpp -- the no-GPU path of BASELINE config 1 ("3-doc README example ... CPU path (plumbing, no GPU)";
// README.md:13-27, suffix_array.pyx:129-180): a small host implementation of the handle API, selected ONLY when no HIP
// device is usable AND the caller has opted in with SA_HIP_ALLOW_HOST=1 (otherwise every entry point keeps failing
// loudly with SA_HIP_EHIP, as before).  It exists so that the package's plumbing -- the Cython class, record retrieval,
// save / load -- can be exercised on a machine without a GPU; it is NOT a fallback of the device path (a process that
// has a device never takes it) and it is deliberately simple: texts of at most HOST_MAX_N bytes.
//
// Own code, unrelated to oracle/ (the test-only checker) and to libsais:
//   construction  prefix doubling over (rank[i], rank[i + h]) pairs with std::sort -- O(n log^2 n), robust on long
//                 repeats; truncated order (engine.c:837-866: first L bytes, ties in text order) ends with ONE step
//                 over (rank_p[i], rank_p[i + L - p]), p the largest power of two <= L: two windows of p characters
//                 cover exactly the first L
//   query         lower / upper bound by memcmp over the suffix array with the conventions of get_substring_positions
//                 (engine.c:869-918; the uint32 wrap at mid == 0 not reproduced)
#pragma once
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "common.hpp"

namespace sa {

constexpr u64 HOST_MAX_N = 1ull << 24;

inline bool host_path_allowed() {
    const char* e = getenv("SA_HIP_ALLOW_HOST");
    return e && atoi(e) != 0;
}

struct HostIndex {
    std::vector<u8> text;
    std::vector<u32> sa;
    u64 n = 0, n_max = 0;
    u32 L = 0;
    u64 freq[256] = {};
    bool ready = false;

    void set_text(const u8* t, u64 n_) {
        n = n_;
        text.assign(t, t + n_);
        memset(freq, 0, sizeof freq);
        for (u64 i = 0; i < n; ++i) ++freq[text[i]];
    }

    // suffixes ordered by their first `limit` bytes (0 = all), a suffix that ends first, ties in text order
    void build(u32 limit) {
        L = limit;
        sa.resize(n);
        std::iota(sa.begin(), sa.end(), 0u);
        ready = true;
        if (n < 2) return;
        std::vector<u32> rank(n), tmp(n);
        for (u64 i = 0; i < n; ++i) rank[i] = (u32)text[i] + 1u;
        auto second = [&](u32 i, u64 h) -> u32 { return (u64)i + h < n ? rank[i + h] : 0u; };
        // one sorting step: order by (rank[i], rank[i + h]), ties by position; returns true when every rank is distinct
        auto step = [&](u64 h) -> bool {
            std::sort(sa.begin(), sa.end(), [&](u32 a, u32 b) {
                if (rank[a] != rank[b]) return rank[a] < rank[b];
                const u32 ra = second(a, h), rb = second(b, h);
                if (ra != rb) return ra < rb;
                return a < b;
            });
            tmp[sa[0]] = 1;
            bool distinct = true;
            for (u64 j = 1; j < n; ++j) {
                const bool same = rank[sa[j]] == rank[sa[j - 1]] && second(sa[j], h) == second(sa[j - 1], h);
                tmp[sa[j]] = tmp[sa[j - 1]] + (same ? 0u : 1u);
                distinct = distinct && !same;
            }
            rank.swap(tmp);
            return distinct;
        };
        if (limit == 1) {   // ordered by one byte, ties in text order
            std::stable_sort(sa.begin(), sa.end(), [&](u32 a, u32 b) { return text[a] < text[b]; });
            return;
        }
        u64 p = 1;          // rank[] orders the first p bytes
        while (true) {
            if (limit && 2 * p > limit) {   // the last, shorter step: windows [0, p) and [limit - p, limit) cover [0, limit)
                if (limit > p) (void)step((u64)limit - p);
                else std::stable_sort(sa.begin(), sa.end(), [&](u32 a, u32 b) { return rank[a] < rank[b]; });
                return;
            }
            const bool distinct = step(p);
            p *= 2;
            if (distinct || p >= n) {
                if (!distinct) (void)step(p);   // p >= n: one more step cannot be needed, but costs nothing to be sure
                return;
            }
        }
    }

    // <0, 0, >0: suffix at pos against the pattern's first c bytes (a suffix that ends before c bytes compares less)
    int cmp(u32 pos, const u8* q, u64 c) const {
        const u64 avail = n - pos;
        const u64 l = avail < c ? avail : c;
        const int r = l ? memcmp(text.data() + pos, q, (size_t)l) : 0;
        if (r) return r;
        return l < c ? -1 : 0;
    }
    sa_hip_pair_u32 query(const u8* q, u64 len) const {
        u64 c = len;
        if (L && c > L) c = L;
        u64 lo = 0, hi = n;
        while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (cmp(sa[mid], q, c) < 0) lo = mid + 1; else hi = mid; }
        const u64 lb = lo;
        hi = n;
        while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (cmp(sa[mid], q, c) <= 0) lo = mid + 1; else hi = mid; }
        sa_hip_pair_u32 r;
        if (lb == n) { r.first = r.second = 0xFFFFFFFFu; }
        else { r.first = (u32)lb; r.second = (u32)(lo - 1); }
        return r;
    }

    // number of violations of the (truncated) suffix-array property
    u64 verify() const {
        u64 bad = 0;
        std::vector<u8> seen(n, 0);
        for (u64 j = 0; j < n; ++j) {
            if (sa[j] >= n || seen[sa[j]]) { ++bad; continue; }
            seen[sa[j]] = 1;
            if (j == 0 || sa[j - 1] >= n) continue;
            const u32 a = sa[j - 1], b = sa[j];
            const u64 la = n - a, lb = n - b;
            u64 c = L ? (u64)L : (la > lb ? la : lb);
            const u64 ca = la < c ? la : c, cb = lb < c ? lb : c;
            const u64 m = ca < cb ? ca : cb;
            int r = m ? memcmp(text.data() + a, text.data() + b, (size_t)m) : 0;
            if (r == 0) r = (ca < cb) ? -1 : (ca > cb ? 1 : 0);
            if (r > 0 || (r == 0 && a > b)) ++bad;
        }
        return bad;
    }
};

}  // namespace sa
