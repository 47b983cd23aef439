// sa_query.hpp -- batched substring search over (text, suffix array) resident in HBM.
//
// Replaces, per element of the batch, get_substring_positions (engine.c:869-918): the inclusive
// SA range of suffixes whose first c = min(len, max_suffix_length) bytes equal the pattern.
// Compare semantics = strncmp over a NUL-terminated text with a NUL-free pattern
// (engine.c:886,906): unsigned bytes; a suffix that ends before c bytes compares less.
// Result conventions (engine.c:896-898, 916-917): {lb, ub-1}; lb == n -> {UINT32_MAX,UINT32_MAX}.
// The reference's uint32 wrap of `mid - 1` at mid == 0 (engine.c:891,908) is NOT reproduced.
//
// The reference probes SA[mid] and then the text at SA[mid]: two dependent random HBM reads per
// step, ~2*log2(n) steps.  Here one lane per query works on three HBM-resident structures:
//   K    u64[n]  the packed first-k0-characters key of every SA slot (the build's sorted key array,
//                kept instead of discarded; b bits per character after alphabet compaction).  A
//                search step is ONE aligned 8-byte load, no indirection, and the last three steps
//                of a descent fall into one 64-byte sector.
//                After a narrow-record sort (keys of <= 40 bits) K is u32[n] instead: (u32)(key >> (64 - key bits)), i.e.
//                the key without (all of) its top 8 bits, which are the same for every slot of a directory bucket (dbits >= 8) and therefore
//                never need to be looked at inside one: 4-byte probes, four steps per sector (query_kernel<true>).
//   dir  u32[2^dbits + 1]  bucket directory over the top dbits of K (first slot of every bucket):
//                one load replaces the top ~24 levels of the descent (it stays resident in the
//                Infinity Cache across the batch).
//   SA,T only for the final disambiguation beyond k0 characters, inside the (usually 0..2 slot)
//                range that K leaves.
// The lower-bound descent remembers the tightest strictly-greater slot, so the upper bound
// starts inside [lb, hi_strict).
#pragma once
#include <type_traits>
#include "common.hpp"
#include "sa_build.hpp"

namespace sa {

__device__ __forceinline__ u64 load_be64(const u8* p) {
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}

// the first 32 bytes of the pattern as four big-endian words, zero padded past c.  Every word that holds pattern bytes is
// loaded whole -- the loads stand together, none waits for another -- and the last partial one is masked (the pattern buffer
// is readable for 8 bytes past its end: sa_hip.h).  Round 3: the byte loop this replaces made every word's load wait for
// the one before it.
__device__ __forceinline__ void pattern_words(const u8* q, u32 c, u64 (&qw)[4]) {
    // no branch around a load (the compiler ends every predicated block with a full wait): words past the pattern re-read its
    // last word and are discarded by a select
    const u32 last = c ? ((c - 1u) & ~7u) : 0u;
    u64 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = load_be64(q + ((u32)(8 * j) < last ? (u32)(8 * j) : last));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 o = 8u * j;
        u64 w = (o < c) ? v[j] : 0ull;
        if (c > o && c - o < 8) w &= ~0ull << (64 - 8 * (c - o));
        qw[j] = w;
    }
}

// three-way compare of suffix `pos` against the pattern (c bytes): <0, 0, >0
template <int WORDS>
__device__ __forceinline__ int cmp_suffix(const u8* __restrict__ text, u64 n, u32 pos, const u64 (&qw)[WORDS],
                                          const u8* __restrict__ q, u32 c) {
    const u64 avail = (pos < n) ? n - pos : 0;   // (an adopted SA is range-checked on load; this keeps a stale entry harmless)
    const u32 l = avail < c ? (u32)avail : c;  // bytes of the suffix that exist
    const u8* s = text + (pos < n ? pos : 0);
    u32 i = 0;
    // whole 8-byte words (text is zero padded, reads past n are in bounds); every word of the register window that holds suffix
    // bytes is requested before the first is compared: the words share a line or two, the waits do not add up
    u64 aw[WORDS];
    const u32 lastw = l ? ((l - 1u) & ~7u) : 0u;   // (no branch around a load: words past the compared length re-read the last one)
#pragma unroll
    for (int w = 0; w < WORDS; ++w) aw[w] = load_be64(s + ((u32)(8 * w) < lastw ? (u32)(8 * w) : lastw));
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
        if (i + 8 <= l) {
            const u64 a = aw[w];
            if (a != qw[w]) return a < qw[w] ? -1 : 1;
            i += 8;
        }
    }
    // further words of patterns longer than the register window, then the 1..7 byte tail
    for (; i + 8 <= l; i += 8) {
        const u64 a = load_be64(s + i), b = load_be64(q + i);
        if (a != b) return a < b ? -1 : 1;
    }
    if (i < l) {
        const u32 r = l - i;
        const u64 m = ~0ull << (64 - 8 * r);
        u64 a, b;
        if (i < (u32)WORDS * 8) {   // inside the register window: i is a multiple of 8 here
            a = 0; b = 0;
#pragma unroll
            for (int w = 0; w < WORDS; ++w) if (i == (u32)(8 * w)) { a = aw[w] & m; b = qw[w] & m; }
        } else { a = load_be64(s + i) & m; b = load_be64(q + i) & m; }
        if (a != b) return a < b ? -1 : 1;
    }
    return l < c ? -1 : 0;
}

struct QueryArgs {
    const u8* text;
    const u32* sa;
    u64 n;
    u32 max_suffix_length;  // 0 = unlimited
    const u8* patterns;     // packed; zero padded by >= 8 readable bytes
    const u64* offsets;     // [q + 1]; nullptr: every pattern has fixed_len bytes, pattern i at patterns + i * fixed_len
    u64 fixed_len;
    u64 q;
    sa_hip_pair_u32* out;
    // acceleration structures (keys == nullptr: plain SA/text descent)
    const u64* keys;        // K
    const u32* dir;         // bucket directory, 2^dbits + 1 entries
    int b, k0, dbits;
    const u32* keys32;      // NARROW: K[j] = (top digit of j's directory bucket << 56) | (keys32[j] << lo_shift)
    int lo_shift;
    int sector_search;      // 1: interpolated sector scan inside the directory bucket (below), 64-byte windows; 2: 32-byte windows; 0: plain binary search
    const u64* keys2;       // second-level keys (k2_build_kernel) or nullptr; wide keys only
    const u64* skeys;       // every SKEY_STRIDE-th key of `keys` (with keys2) or nullptr
    const u32* perm;        // or nullptr: thread i answers query perm[i] (a large batch over a wide-key index, clustered by its patterns' first characters)
    int k2n;                // characters a second-level key holds
};

// ---- second-level keys (round 4) ---------------------------------------------------------------------------------------------
// A pattern longer than the key (k0 characters) leaves the slots that share its first k0 characters; the bounds inside that
// range used to be found by comparing TEXT -- SA[mid], then the suffix there: two dependent random reads per step, and on
// name-like text (every pattern a hit, popular prefixes shared by thousands of suffixes) that search was the whole batch:
// 0.55 ms per 1e6 names, ~28 requests per query.  K2[j] = the k2n characters that follow the key of slot j (codes of b bits,
// MSB first, left aligned; 0 past the end of the text or of a truncated index's depth), for every slot whose key equals a
// neighbour's -- only those are ever inside a range of more than one slot.  Inside a key group the slots are in suffix order,
// so K2 is sorted there: the bounds are two binary searches over 8-byte keys (one read per step, the last steps in one
// sector), exact for patterns of up to k0 + k2n characters (23 at 5 bits), a narrower range for the text search beyond.
__global__ __launch_bounds__(256) void k2_build_kernel(const u64* __restrict__ K, const u32* __restrict__ sa, const u8* __restrict__ text, u64 n,
                                                       CodeMap map, int b, int k0, int k2n, u32 L, u64* __restrict__ K2) {
    __shared__ u16 s_map[256];
    s_map[threadIdx.x] = map.code[threadIdx.x];
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u32 depth = (L && L > (u32)k0) ? L - (u32)k0 : (L ? 0u : 0xFFFFFFFFu);   // characters beyond a truncated index's depth do not order anything
    // a key group = the slots that share the key's first k0 CHARACTERS -- what a pattern's key range spans; the stored key may
    // hold bits beyond them (10-byte-record plan: 55 bits of characters + one bit of the twelfth), which do not count here
    const int gs = 64 - k0 * b;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const u64 k = K[j] >> gs;
        const bool tied = (j > 0 && (K[j - 1] >> gs) == k) || (j + 1 < n && (K[j + 1] >> gs) == k);
        if (!tied) continue;   // (never read: a slot alone in its key group is a range of one)
        const u64 pos = (u64)sa[j] + (u64)k0;
        u64 w[2] = {0, 0};
        if (pos < n) { __builtin_memcpy(&w[0], text + pos, 8); __builtin_memcpy(&w[1], text + pos + 8, 8); }   // the text is zero padded: in bounds
        u64 avail = pos < n ? n - pos : 0;
        if (avail > depth) avail = depth;
        u64 key = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < k2n) {   // uniform
                const u32 byte = (u32)(w[c >> 3] >> (8 * (c & 7))) & 255u;
                key = (key << b) | (((u64)c < avail) ? (u64)s_map[byte] : 0ull);
            }
        }
        K2[j] = key << (64 - k2n * b);
    }
}

// The batch is bound by the NUMBER of 64-byte sectors it requests, not by bytes or by the dependent-load chain
// (profiles/r02_a_pmc_*: 5.5 requests per query at the ~36 G requests/s the memory system sustains for random reads,
// tools/gatherbench).  A binary search over the ~35 slots of a directory bucket touches 2-3 sectors of K.  The keys of
// a bucket are close to uniformly spread over the bucket's key interval, so the place of a key is first ESTIMATED by
// linear interpolation and the whole sector around it is read (one request: the lane's four 16-byte loads hit the line
// the first one brings in); sorted keys tell at once whether the bound lies inside, to the left or to the right.
// Returns the first slot in [l, h) whose key is >= t (h if none).  win / wbase: the keys of the sector the answer was
// found in and the slot of win[0] (for the upper bound, which usually lies in the same sector).
// WBYTES: 64 (a whole sector) or 32 (half of one).  Round 3 (tools/gatherbench sweep, profiles/r03_gather_sweep.log): random
// reads from a footprint beyond the Infinity Cache are served at ~1.65 TB/s in 32-BYTE granules whatever the request size
// (4-byte reads 51 G/s, 64-byte reads 26 G/s, 128-byte reads 13 G/s) -- a 64-byte window costs two granules, a 32-byte
// window one, and the interpolated estimate is usually within the 8 (u32) keys around it.
template <typename KT, int WBYTES>
struct SectorWindow {
    static constexpr int W = WBYTES / (int)sizeof(KT);
    KT k[W];
    u64 base;    // slot of k[0] (a multiple of W)
};
template <typename KT, int WBYTES>
__device__ __forceinline__ void load_sector(const KT* __restrict__ K, u64 base, SectorWindow<KT, WBYTES>& w) {
    w.base = base;
    const uint4* p = reinterpret_cast<const uint4*>(K + base);   // K is 256-byte aligned and padded to whole sectors
#pragma unroll
    for (int i = 0; i < WBYTES / 16; ++i) {
        const uint4 v = p[i];
        if (sizeof(KT) == 4) { w.k[4 * i] = (KT)v.x; w.k[4 * i + 1] = (KT)v.y; w.k[4 * i + 2] = (KT)v.z; w.k[4 * i + 3] = (KT)v.w; }
        else { w.k[2 * i] = (KT)(((u64)v.y << 32) | v.x); w.k[2 * i + 1] = (KT)(((u64)v.w << 32) | v.z); }
    }
}
// first slot in [l, h) with key >= t (STRICT: > t), starting at the sector of `est`; l < h
// S (wide keys, with the second-level keys; nullptr otherwise): every SKEY_STRIDE-th key of K.  When the windows around the
// estimate do not hold the bound -- on word / name text a directory bucket (the first 5.4 characters) can hold 10^7 slots and the
// keys inside it are anything but uniformly spread -- the bisection runs over the SAMPLES first: log2(range / 256) reads of
// an array that stays in the caches (116 MB at config-5 size with every 64th key), then at most 6 steps in K itself, instead of log2(range) reads
// of K from DRAM.
#ifndef SA_SKEY_STRIDE
#define SA_SKEY_STRIDE 64
#endif
constexpr u64 SKEY_STRIDE = SA_SKEY_STRIDE;
template <bool STRICT, typename KT, int WBYTES>
__device__ __forceinline__ u64 sector_bound(const KT* __restrict__ K, u64 l, u64 h, KT t, u64 est, SectorWindow<KT, WBYTES>& w, bool have_window,
                                            const KT* __restrict__ S = nullptr, int window_steps = 3) {
    constexpr int W = SectorWindow<KT, WBYTES>::W;
    u64 lo = l, hi = h;   // the bound lies in [lo, hi]
    u64 base = est & ~(u64)(W - 1);
    for (int step = 0; step < window_steps && lo < hi; ++step) {
        if (!(have_window && w.base == base)) load_sector(K, base, w);
        have_window = false;
        // slots of this sector inside [lo, hi)
        const u64 a = base > lo ? base : lo;
        const u64 b = (base + W < hi) ? base + W : hi;
        u32 below = 0;   // keys of [a, b) that are < t (<= t)
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const u64 slot = base + i;
            const bool in = slot >= a && slot < b;
            const bool less = STRICT ? (w.k[i] <= t) : (w.k[i] < t);
            below += (in && less) ? 1u : 0u;
        }
        if (below == 0) {             // every key here is >= t: the bound is a or further left
            hi = a;
            if (a == lo) break;
            base -= W;                // (a > lo: a == base, the sector before exists)
        } else if (below == (u32)(b - a)) {   // every key here is < t: further right
            lo = b;
            if (b == hi) break;
            base += W;
        } else { lo = hi = a + below; }
    }
    if (S && hi - lo > 2 * SKEY_STRIDE) {
        // first sample index i in [ia, ib) whose key is >= t (> t); sample i is slot i * SKEY_STRIDE, all of them inside [lo, hi)
        u64 ia = (lo + SKEY_STRIDE - 1) / SKEY_STRIDE, ib = (hi - 1) / SKEY_STRIDE + 1;
        const u64 ia0 = ia, ib0 = ib;
        while (ia < ib) {
            const u64 mid = (ia + ib) >> 1;
            const KT k = S[mid];
            if (STRICT ? (k <= t) : (k < t)) ia = mid + 1; else ib = mid;
        }
        if (ia < ib0) hi = ia * SKEY_STRIDE;             // that slot satisfies the bound's condition: the bound is at or before it
        if (ia > ia0) lo = (ia - 1) * SKEY_STRIDE + 1;   // the sample before does not: the bound is after it
    }
    while (lo < hi) {   // (narrow keys: rare -- the estimate was more than two sectors off: long runs of equal keys, lumpy buckets)
        const u64 mid = (lo + hi) >> 1;
        const KT k = K[mid];
        if (STRICT ? (k <= t) : (k < t)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void skeys_kernel(const u64* __restrict__ K, u64 n, u64* __restrict__ S) {
    const u64 ns = (n + SKEY_STRIDE - 1) / SKEY_STRIDE;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += stride) S[i] = K[i * SKEY_STRIDE];
}

// MODE = QueryArgs::sector_search as a compile-time constant: the product path (2) does not carry the registers of the
// 64-byte windows or of the plain bisection (diagnostic modes 1 and 0, kept for the tests)
// one query: the inclusive SA range of pattern qi (conventions of get_substring_positions, engine.c:896-898, 916-917)
template <bool NARROW, int MODE>
__device__ __forceinline__ sa_hip_pair_u32 query_one(const QueryArgs& a, const u16* s_map, const u64 qi) {
    using KT = typename std::conditional<NARROW, u32, u64>::type;
    const KT* __restrict__ K = NARROW ? reinterpret_cast<const KT*>(a.keys32) : reinterpret_cast<const KT*>(a.keys);
    {
        u64 o = qi * a.fixed_len, len = a.fixed_len;
        if (a.offsets) {   // (uniform branch; both loads requested together)
            const u64 o0 = a.offsets[qi], o1 = a.offsets[qi + 1];
            o = o0; len = o1 - o0;
        }
        u32 c = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)len;
        if (a.max_suffix_length && c > a.max_suffix_length) c = a.max_suffix_length;
        const u8* q = a.patterns + o;
        u64 qw[4];
        pattern_words(q, c, qw);

        // ---- phase 1: narrow to the slots whose first P characters equal the pattern's -----------
        u64 lo = 0, hi = a.n;      // lb and ub both lie in [lo, hi]
        bool exact = false;        // the K range IS the answer (whole pattern packed)
        int P = 0;                 // pattern characters the key search has matched
        if (K) {
            u64 key_lo = 0;
            int sh = 64;
            const int pmax = (c < (u32)a.k0) ? (int)c : a.k0;
            // the pattern's characters as codes, four look-ups in flight at a time (one at a time waited an LDS round trip
            // per character); a byte absent from the text (code 0) ends the key: nothing matches past it
            bool stop = false;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (4 * g < pmax && !stop) {
                    const u32 w32 = (g & 1) ? (u32)qw[g >> 1] : (u32)(qw[g >> 1] >> 32);
                    const u32 cd[4] = {s_map[w32 >> 24], s_map[(w32 >> 16) & 255u], s_map[(w32 >> 8) & 255u], s_map[w32 & 255u]};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (P < pmax && !stop) {
                            if (cd[t] == 0) stop = true;
                            else { sh -= a.b; key_lo |= (u64)cd[t] << sh; ++P; }
                        }
                    }
                }
            }
            for (; P < pmax && !stop; ++P) {   // keys of more than 32 characters (one-bit alphabets)
                const u32 code = s_map[q[P]];
                if (code == 0) break;
                sh -= a.b;
                key_lo |= (u64)code << sh;
            }
            if (P > 0) {
                const u64 key_hi = key_lo | ((sh > 0) ? ((1ull << sh) - 1ull) : 0ull);
                const int ds = 64 - a.dbits;
                const u32 bl = (u32)(key_lo >> ds), bh = (u32)(key_hi >> ds);
                u64 l = a.dir[bl], h = a.dir[bl + 1];
                u64 h_strict = (bh == bl) ? h : a.dir[bh + 1];   // first slot known to be > key_hi
                const u64 h2_lo = (bh == bl) ? 0 : a.dir[bh];
                // thresholds in the domain of the stored keys.  NARROW: every slot of a directory bucket has the
                // bucket's top 8 key bits (dbits >= 8), so inside bucket bl "K < key_lo" is a comparison of the
                // bits below them, and "K > key_hi" can only hold when key_hi has the same top digit; the bits
                // below lo_shift are zero in K and in key_lo and all ones in key_hi (sh >= lo_shift).
                KT t_lo, t_hi1, t_hi2;
                bool searched = false;
                if (NARROW) {
                    // the stored form is (u32)(key >> lo_shift): keys of 40 bits lose their top digit in the
                    // truncation, shorter ones keep (part of) it -- the thresholds are truncated the same way
                    t_lo = (KT)(key_lo >> a.lo_shift);
                    t_hi2 = (KT)(key_hi >> a.lo_shift);
                    t_hi1 = ((key_lo >> 56) == (key_hi >> 56)) ? t_hi2 : (KT)~(KT)0;
                } else { t_lo = (KT)key_lo; t_hi1 = t_hi2 = (KT)key_hi; }
                if (MODE != 0 && bh == bl && l < h) {
                    // one bucket: its keys lie in [bkt << ds, (bkt + 1) << ds); estimate the place of key_lo by its
                    // position in that interval (stored form: bits below the bucket bits, above lo_shift)
                    const int vs = NARROW ? a.lo_shift : 0;                         // low bit of the stored form
                    const int sb = ds - vs;                                         // stored bits below the bucket bits
                    u64 est = l;
                    if (sb > 0) {
                        const u64 frac = (key_lo >> vs) & ((1ull << sb) - 1ull);    // position inside the bucket's interval
                        const int down = sb > 20 ? sb - 20 : 0;                     // 20 significant bits are plenty for <= 2^20 slots
                        est = l + (((frac >> down) * (h - l)) >> (sb - down));
                        if (est >= h) est = h - 1;
                    }
                    const KT* S = NARROW ? (const KT*)nullptr : reinterpret_cast<const KT*>(a.skeys);
                    if (MODE == 2) {   // 32-byte windows
                        SectorWindow<KT, 32> w;
                        // with sampled keys and a bucket far larger than a window the interpolated estimate is not worth its three
                        // requests (text is not uniformly spread inside a bucket): samples first; the upper bound still looks at the
                        // window the lower bound ended in (a small group ends there) and one more
                        const bool big = S && h - l > 64 * SKEY_STRIDE;
                        lo = sector_bound<false>(K, l, h, t_lo, est, w, false, S, big ? 0 : 3);
                        hi = (lo < h) ? sector_bound<true>(K, lo, h, t_hi2, lo, w, !big, S, big ? 2 : 3) : lo;
                    } else {
                        SectorWindow<KT, 64> w;
                        lo = sector_bound<false>(K, l, h, t_lo, est, w, false, S);
                        hi = (lo < h) ? sector_bound<true>(K, lo, h, t_hi2, lo, w, true, S) : lo;
                    }
                    searched = true;
                }
                // first slot with K >= key_lo
                while (!searched && l < h) {
                    const u64 mid = (l + h) >> 1;
                    const KT k = K[mid];
                    if (k < t_lo) l = mid + 1;
                    else { h = mid; if (k > t_hi1) h_strict = mid; }
                }
                if (!searched) {
                lo = l;
                // first slot with K > key_hi, inside [max(lo, dir[bh]), h_strict]
                u64 l2 = (bh == bl) ? lo : (h2_lo > lo ? h2_lo : lo);
                u64 h2 = h_strict;
                while (l2 < h2) {
                    const u64 mid = (l2 + h2) >> 1;
                    if (K[mid] <= t_hi2) l2 = mid + 1; else h2 = mid;
                }
                hi = l2;
                }
                exact = ((u32)P == c);
            }
        }

        // ---- phase 1b: second-level keys inside a key group of more than one slot (wide keys, when they have been built)
        if (!NARROW && a.keys2 && !exact && hi - lo > 1 && P == a.k0) {
            const int m2 = ((int)(c - (u32)a.k0) < a.k2n) ? (int)(c - (u32)a.k0) : a.k2n;   // >= 1: the pattern is longer than the key
            u64 q2 = 0;
            bool absent = false;   // a byte that does not occur in the text: nothing matches; the text search below says where
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i < m2) {
                    const u32 x = (u32)a.k0 + (u32)i;
                    u32 byte;
                    if (x < 32u) {
                        const u64 wsel = (x < 8u) ? qw[0] : (x < 16u) ? qw[1] : (x < 24u) ? qw[2] : qw[3];
                        byte = (u32)(wsel >> (56u - 8u * (x & 7u))) & 255u;
                    } else byte = q[x];
                    const u32 code = s_map[byte];
                    absent |= (code == 0);
                    q2 = (q2 << a.b) | code;
                }
            }
            if (!absent) {
                const int sh2 = 64 - m2 * a.b;   // 4 <= sh2 < 64
                const u64 q2_lo = q2 << sh2, q2_hi = q2_lo | ((1ull << sh2) - 1ull);
                u64 l = lo, h = hi, h_strict = hi;   // the lower-bound descent remembers the tightest slot known to lie above the pattern
                while (l < h) {
                    const u64 mid = (l + h) >> 1;
                    const u64 k2 = a.keys2[mid];
                    if (k2 < q2_lo) l = mid + 1;
                    else { h = mid; if (k2 > q2_hi) h_strict = mid; }
                }
                const u64 lb2 = l;
                h = h_strict;
                while (l < h) {
                    const u64 mid = (l + h) >> 1;
                    if (a.keys2[mid] <= q2_hi) l = mid + 1; else h = mid;
                }
                lo = lb2; hi = l;
                exact = (c <= (u32)(a.k0 + a.k2n));
            }
        }

        // ---- phase 2: text comparisons inside [lo, hi) ---------------------------------------------
        u64 lb = lo, ub = hi;
        if (!exact && hi - lo == 1) {
            // one candidate slot (the usual case once K has resolved the first k0 characters): one compare
            // settles both bounds
            const int r = cmp_suffix<4>(a.text, a.n, a.sa[lo], qw, q, c);
            lb = (r < 0) ? hi : lo;
            ub = (r > 0) ? lo : hi;
        } else if (!exact) {
            u64 l = lo, h = hi, h_strict = hi;
            while (l < h) {
                const u64 mid = (l + h) >> 1;
                const int r = cmp_suffix<4>(a.text, a.n, a.sa[mid], qw, q, c);
                if (r < 0) l = mid + 1;
                else { h = mid; if (r > 0) h_strict = mid; }
            }
            lb = l;
            h = h_strict;
            while (l < h) {
                const u64 mid = (l + h) >> 1;
                const int r = cmp_suffix<4>(a.text, a.n, a.sa[mid], qw, q, c);
                if (r <= 0) l = mid + 1;
                else h = mid;
            }
            ub = l;
        }
        sa_hip_pair_u32 res;
        if (lb == a.n) { res.first = 0xFFFFFFFFu; res.second = 0xFFFFFFFFu; }
        else { res.first = (u32)lb; res.second = (u32)(ub - 1); }
        return res;
    }
}

template <bool NARROW, int MODE = 2>
__global__ __launch_bounds__(256) void query_kernel(QueryArgs a, CodeMap map) {   // MODE 2: 72 registers, 7 waves per SIMD (forced to 8: 0.0765 instead of 0.0734 ms)
    __shared__ u16 s_map[256];
    s_map[threadIdx.x] = map.code[threadIdx.x];
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 qi = (u64)blockIdx.x * blockDim.x + threadIdx.x; qi < a.q; qi += stride) {
        const u64 qq = (!NARROW && a.perm) ? (u64)a.perm[qi] : qi;
        a.out[qq] = query_one<NARROW, MODE>(a, s_map, qq);
    }
}

// ---- clustering a large batch (round 4) ----------------------------------------------------------------------------------------
// On word / name text the patterns of a batch walk key groups of very different sizes (a popular prefix is shared by 10^6
// suffixes, most by a handful): a wave runs as long as its slowest lane, and neighbouring lanes touch unrelated cache lines.
// The same 1e6 names in lexicographic order were answered in 0.170 instead of 0.330 ms.  So a batch of >= 32768 patterns over a
// wide-key index is answered in the order of its patterns' first characters (22 bits of the packed codes: 4.4 characters at 5
// bits): two counting passes of 11 bits over tiles of 4096 queries -- the place inside a tile's bin from an LDS atomic (any order
// inside a bin will do: the passes need not be stable, the order only has to bring like patterns together), the tiles' counts
// scanned bin-major -- then thread i answers query perm[i] and writes out[perm[i]].  (A first form counted with returning
// atomics on global counters: 0.30 ms for that pass alone -- device-scope atomics are served on the memory side, not in an XCD's
// L2 -- whatever the number of counters.)
constexpr int QC_DIGIT_BITS = 11;
constexpr u32 QC_BINS = 1u << QC_DIGIT_BITS;
constexpr u32 QC_TILE = 4096;
// pass 0: the keys from the patterns (kept in qkey), ranked by their LOW digit; pass 1: by the high digit, in pass 0's order
__global__ __launch_bounds__(256) void qcluster_rank_kernel(QueryArgs a, CodeMap map, int pass, const u32* __restrict__ order, u32* __restrict__ qkey,
                                                            u32* __restrict__ tmp, u32* __restrict__ tile_hist, u32 ntiles) {
    __shared__ u16 s_map[256];
    __shared__ u32 s_cnt[QC_BINS];
    s_map[threadIdx.x] = map.code[threadIdx.x];
    for (u32 i = threadIdx.x; i < QC_BINS; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * QC_TILE;
    for (u32 e = 0; e < QC_TILE / 256; ++e) {
        const u64 i = base + (u64)e * 256 + threadIdx.x;
        if (i >= a.q) break;
        u32 key;
        if (pass == 0) {
            u64 o = i * a.fixed_len, len = a.fixed_len;
            if (a.offsets) { const u64 o0 = a.offsets[i], o1 = a.offsets[i + 1]; o = o0; len = o1 - o0; }
            const u64 w = load_be64(a.patterns + o);   // (the pattern buffer is readable for 8 bytes past its end)
            u64 k = 0;
            int bits = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (bits < 2 * QC_DIGIT_BITS) {   // (uniform)
                    const u32 code = ((u64)c < len) ? (u32)s_map[(u32)(w >> (56 - 8 * c)) & 255u] : 0u;
                    k = (k << a.b) | code;
                    bits += a.b;
                }
            }
            key = (u32)(bits >= 2 * QC_DIGIT_BITS ? k >> (bits - 2 * QC_DIGIT_BITS) : k << (2 * QC_DIGIT_BITS - bits));
            qkey[i] = key;
        } else key = qkey[order[i]];
        const u32 d = pass == 0 ? (key & (QC_BINS - 1u)) : (key >> QC_DIGIT_BITS);
        tmp[i] = (atomicAdd(&s_cnt[d], 1u) << QC_DIGIT_BITS) | d;
    }
    sync_lds();
    for (u32 d = threadIdx.x; d < QC_BINS; d += 256) tile_hist[(u64)d * ntiles + blockIdx.x] = s_cnt[d];
}
__global__ __launch_bounds__(256) void qcluster_place_kernel(u64 q, const u64* __restrict__ off, const u32* __restrict__ tmp, const u32* __restrict__ order,
                                                             u32 ntiles, u32* __restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += stride) {
        const u32 t = tmp[i], d = t & (QC_BINS - 1u), r = t >> QC_DIGIT_BITS;
        out[off[(u64)d * ntiles + i / QC_TILE] + r] = order ? order[i] : (u32)i;
    }
}

}  // namespace sa
