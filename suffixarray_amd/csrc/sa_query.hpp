// sa_query.hpp -- batched substring search over (text, suffix array) resident in HBM.
//
// Replaces, per element of the batch, get_substring_positions (engine.c:869-918): the inclusive
// SA range of suffixes whose first c = min(len, max_suffix_length) bytes equal the pattern.
// Compare semantics = strncmp over a NUL-terminated text with a NUL-free pattern
// (engine.c:886,906): unsigned bytes; a suffix that ends before c bytes compares less.
// Result conventions (engine.c:896-898, 916-917): {lb, ub-1}; lb == n -> {UINT32_MAX,UINT32_MAX}.
// The reference's uint32 wrap of `mid - 1` at mid == 0 (engine.c:891,908) is NOT reproduced.
//
// v1 kernel: one lane per query.  The lower-bound descent remembers the tightest strictly
// greater slot, so the upper-bound search starts inside [lb, hi_strict) -- for a miss it
// costs one extra probe instead of a second full descent.  Pattern bytes are fetched once into
// registers as big-endian 64-bit words (<= 32 bytes; longer patterns fall back to memory).
#pragma once
#include "common.hpp"

namespace sa {

__device__ __forceinline__ u64 load_be64(const u8* p) {
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}

// big-endian word j of the pattern, zero padded past c
__device__ __forceinline__ u64 pattern_word(const u8* q, u32 c, u32 j) {
    u64 v = 0;
    const u32 o = j * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u32 i = o + k;
        v = (v << 8) | (u64)((i < c) ? q[i] : 0);
    }
    return v;
}

// three-way compare of suffix `pos` against the pattern (c bytes): <0, 0, >0
template <int WORDS>
__device__ __forceinline__ int cmp_suffix(const u8* __restrict__ text, u64 n, u32 pos, const u64 (&qw)[WORDS],
                                          const u8* __restrict__ q, u32 c) {
    const u64 avail = n - pos;
    const u32 l = avail < c ? (u32)avail : c;  // bytes of the suffix that exist
    const u8* s = text + pos;
    u32 i = 0;
    // whole 8-byte words (text is zero padded, reads past n are in bounds)
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
        if (i + 8 <= l) {
            const u64 a = load_be64(s + i);
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
        const u64 a = load_be64(s + i) & m;
        const u64 b = (i < (u32)WORDS * 8) ? (qw[i / 8] & m) : (load_be64(q + i) & m);
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
    const u64* offsets;     // [q + 1]
    u64 q;
    sa_hip_pair_u32* out;
};

__global__ __launch_bounds__(256) void query_kernel(QueryArgs a) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 qi = (u64)blockIdx.x * blockDim.x + threadIdx.x; qi < a.q; qi += stride) {
        const u64 o = a.offsets[qi];
        const u64 len = a.offsets[qi + 1] - o;
        u32 c = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)len;
        if (a.max_suffix_length && c > a.max_suffix_length) c = a.max_suffix_length;
        const u8* q = a.patterns + o;
        u64 qw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) qw[j] = pattern_word(q, c, j);

        // lower bound: first slot whose suffix is >= pattern
        u64 lo = 0, hi = a.n, hi_strict = a.n;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            const int r = cmp_suffix<4>(a.text, a.n, a.sa[mid], qw, q, c);
            if (r < 0) lo = mid + 1;
            else { hi = mid; if (r > 0) hi_strict = mid; }
        }
        const u64 lb = lo;
        // upper bound: first slot whose suffix is > pattern, inside [lb, hi_strict]
        hi = hi_strict;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            const int r = cmp_suffix<4>(a.text, a.n, a.sa[mid], qw, q, c);
            if (r <= 0) lo = mid + 1;
            else hi = mid;
        }
        sa_hip_pair_u32 res;
        if (lb == a.n) { res.first = 0xFFFFFFFFu; res.second = 0xFFFFFFFFu; }
        else { res.first = (u32)lb; res.second = (u32)(lo - 1); }
        a.out[qi] = res;
    }
}

}  // namespace sa
