// period_finish.hpp -- tied groups that are ARITHMETIC PROGRESSIONS inside one periodic run are ordered in one step.
//
// Long repeats are what prefix doubling is worst at: a text made of a block of P characters repeated r times leaves,
// after the initial sort, n / r groups {i, i + P, ..., i + (r - 1) P}; every doubling round peels off only the members
// within reach of the text's end, ~log2(n / k0) rounds over ALL n suffixes (1 MiB x 95: 22 rounds, 164 ms against 2.5 ms
// for a random text of the same length; all-'a', period-k and Fibonacci strings alike).  libsais handles the same
// inputs in linear time (induced sorting, libsais.c:6480-6519); this is the corresponding shortcut here.
//
// Members of a group are kept in ascending text position (every sort is stable and starts from the identity).  If a
// group's positions are p0, p0 + d, ..., p0 + (m - 1) d and the text is periodic with period d from p0 up to E -- E the
// first x >= p0 with x + d >= n or T[x] != T[x + d] -- and E >= p0 + (m - 2) d, then ANY two neighbours a = p, b = p + d
// agree on their first E - p characters and differ right there: a has T[E], b has T[E + d] (or has ended).  The outcome
// is the same for every pair of the group, so the whole group is ordered by ONE comparison: descending positions when
// b < a (b ended, or T[E + d] < T[E]), ascending otherwise.  No assumption on how d compares with the depth h.
//
// One attempt = classify the groups (arithmetic? difference d), pick the difference that covers the most records (hash
// histogram), and -- if it covers enough of the active set -- build E(x) for that d over the whole text with three
// streaming kernels (mismatch flags, first mismatch per tile, suffix scan of the tiles, next mismatch per position),
// resolve its groups, repeat for the next difference (Fibonacci-like texts have a few).  Full suffix arrays only (a
// truncated order compares L characters, not to the end of the run).
#pragma once
#include "common.hpp"

namespace sa {

constexpr u32 PER_TABLE = 1u << 16;   // hash histogram of the differences: {d, records} pairs
constexpr int PER_TILE = 4096;        // positions per workgroup of the next-mismatch kernels (256 threads x 16)

// gd[g] = difference of the group's first two members (groups of the active list have >= 2 members); bad[g] = 0
__global__ __launch_bounds__(256) void per_init_kernel(const u32* __restrict__ aidx, const u32* __restrict__ gstart, u32 G,
                                                       u32* __restrict__ gd, u8* __restrict__ bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
        const u32 s = gstart[g];
        const bool pair = s + 1 < gstart[g + 1];   // (groups of the active list have >= 2 members; a singleton would never be touched)
        gd[g] = pair ? aidx[s + 1] - aidx[s] : 0u;
        bad[g] = pair ? 0 : 1;
    }
}
// bad[g] = 1 when some neighbouring members differ by something else
__global__ __launch_bounds__(256) void per_classify_kernel(const u32* __restrict__ aidx, const u32* __restrict__ gid,
                                                           const u32* __restrict__ gstart, u32 M, const u32* __restrict__ gd,
                                                           u8* __restrict__ bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
        const u32 g = gid[j];
        if (j >= (u64)gstart[g] + 2 && aidx[j] - aidx[j - 1] != gd[g]) bad[g] = 1;
    }
}
// table[slot] = {d, records of arithmetic groups with that difference} (open addressing; d >= 1).  A periodic text has ONE
// difference for a million groups: the lanes of a wave that hold the same d are combined first (one atomic per distinct d and
// wave; a million atomics on one word took ~20 ms per call)
__global__ __launch_bounds__(256) void per_hist_kernel(const u32* __restrict__ gstart, u32 G, const u32* __restrict__ gd,
                                                       const u8* __restrict__ bad, uint2* __restrict__ table) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 g0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 rounds = ((u64)G + stride - 1) / stride;
    for (u64 it = 0; it < rounds; ++it) {
        const u64 g = g0 + it * stride;
        u32 d = 0, size = 0;
        if (g < G && !bad[g]) { d = gd[g]; size = gstart[g + 1] - gstart[g]; }
        u64 todo = __ballot(d != 0);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const u32 dl = __shfl(d, leader);
            const u64 same = __ballot(d == dl) & todo;
            u32 sum = (d == dl) ? size : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if ((int)(threadIdx.x & 63) == leader) {
                u32 slot = (dl * 0x9E3779B1u) >> 16;
                for (u32 probe = 0; probe < 64; ++probe) {
                    const u32 old = atomicCAS(&table[slot].x, 0u, dl);
                    if (old == 0u || old == dl) { atomicAdd(&table[slot].y, sum); break; }
                    slot = (slot + 1u) & (PER_TABLE - 1u);
                }
            }
            todo &= ~same;
        }
    }
}
// best[0] = difference with the most records, best[1] = that count (one workgroup)
__global__ __launch_bounds__(1024) void per_pick_kernel(const uint2* __restrict__ table, u32* __restrict__ best) {
    __shared__ u32 s_c[1024], s_d[1024];
    u32 bc = 0, bd = 0;
    for (u32 i = threadIdx.x; i < PER_TABLE; i += 1024) {
        const uint2 e = table[i];
        if (e.y > bc || (e.y == bc && e.y && e.x < bd)) { bc = e.y; bd = e.x; }
    }
    s_c[threadIdx.x] = bc; s_d[threadIdx.x] = bd;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const u32 c2 = s_c[threadIdx.x + o], d2 = s_d[threadIdx.x + o];
            if (c2 > s_c[threadIdx.x] || (c2 == s_c[threadIdx.x] && c2 && d2 < s_d[threadIdx.x])) { s_c[threadIdx.x] = c2; s_d[threadIdx.x] = d2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { best[0] = s_d[0]; best[1] = s_c[0]; }
}

// ---- E(x) = first y >= x with y + d >= n or T[y] != T[y + d], for every x ------------------------------------------
__device__ __forceinline__ bool per_mismatch(const u8* __restrict__ text, u64 n, u64 d, u64 y) {
    return y + d >= n || text[y] != text[y + d];
}
// first mismatch inside every tile (NONE32 when there is none)
__global__ __launch_bounds__(256) void per_tile_first_kernel(const u8* __restrict__ text, u64 n, u64 d, u32* __restrict__ tile_first) {
    __shared__ u32 s_m[4];
    const u64 base = (u64)blockIdx.x * PER_TILE;
    u32 first = 0xFFFFFFFFu;
    for (int it = 0; it < PER_TILE / 256; ++it) {
        const u64 y = base + (u64)it * 256 + threadIdx.x;
        if (y < n && first == 0xFFFFFFFFu && per_mismatch(text, n, d, y)) first = (u32)y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u32 t = __shfl_down(first, o); first = t < first ? t : first; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = first;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 m = s_m[0];
        for (int w = 1; w < 4; ++w) m = s_m[w] < m ? s_m[w] : m;
        tile_first[blockIdx.x] = m;
    }
}
// carry[t] = first mismatch in any tile > t (suffix minimum over the tiles; one workgroup)
__global__ __launch_bounds__(1024) void per_tile_scan_kernel(const u32* __restrict__ tile_first, u32 ntiles, u32* __restrict__ carry) {
    __shared__ u32 s_v[1024];
    const u32 per = (ntiles + 1023) / 1024;
    const u32 lo = threadIdx.x * per;
    const u32 hi = (lo + per < ntiles) ? lo + per : ntiles;
    u32 v = 0xFFFFFFFFu;
    for (u32 i = lo; i < hi; ++i) { const u32 t = tile_first[i]; v = t < v ? t : v; }
    s_v[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // inclusive suffix minimum
        u32 t = 0xFFFFFFFFu;
        if ((int)threadIdx.x + o < 1024) t = s_v[threadIdx.x + o];
        __syncthreads();
        if (t < s_v[threadIdx.x]) s_v[threadIdx.x] = t;
        __syncthreads();
    }
    u32 c = (threadIdx.x + 1 < 1024) ? s_v[threadIdx.x + 1] : 0xFFFFFFFFu;   // everything right of this thread's range
    for (u32 i = hi; i-- > lo;) {
        carry[i] = c;
        const u32 t = tile_first[i];
        c = t < c ? t : c;
    }
}
// Decision per arithmetic group with difference d (one wave per group): E = first mismatch at or after its first member --
// the tile's first mismatch when that lies at or beyond p0, else a forward scan of the rest of p0's tile (64 positions per
// step), else the carry of the tiles to the right.  dec[g] = 1 (ascending positions) / 2 (descending) when the run covers
// every member, 0 otherwise (the run ends inside the group: left to the rounds).
struct PerArgs {
    const u8* text; u64 n; u64 d;
    const u32* aidx; const u32* apos; const u32* gid; const u32* gstart; const u32* gd; u8* bad; u32 G; u32 M;
    const u32* tile_first; const u32* carry;
    u8* dec;
    u32* sa; u8* gflags; u8* done; u32* isa;   // isa may be null
};
__global__ __launch_bounds__(256) void per_decide_kernel(PerArgs a) {
    const int lane = threadIdx.x & 63;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 g = wave; g < a.G; g += nwaves) {
        u8 dec = 0;
        if (!a.bad[g] && (u64)a.gd[g] == a.d) {   // wave-uniform
            const u32 gs = a.gstart[g], ge = a.gstart[g + 1];
            const u64 p0 = a.aidx[gs], plast = a.aidx[ge - 1];
            const u64 t = p0 / PER_TILE, tile_end = (t + 1) * PER_TILE;
            const u32 tf = a.tile_first[t];
            u64 E = ~0ull;
            if (tf != 0xFFFFFFFFu && (u64)tf >= p0) E = tf;
            else if (tf != 0xFFFFFFFFu) {
                for (u64 y0 = p0; y0 < tile_end && E == ~0ull; y0 += 64) {
                    const u64 y = y0 + lane;
                    const bool mm = y < tile_end && y < a.n && per_mismatch(a.text, a.n, a.d, y);
                    const u64 b = __ballot(mm);
                    if (b) E = y0 + (u64)__builtin_ctzll(b);
                }
            }
            if (E == ~0ull) { const u32 c = a.carry[t]; E = (c == 0xFFFFFFFFu) ? a.n : (u64)c; }
            // every neighbouring pair (p_j, p_j + d), j <= m - 2, must meet its first mismatch at the same E: E >= p_{m-2} = plast - d
            if (E + a.d >= plast && E < a.n) dec = ((E + a.d >= a.n) || a.text[E + a.d] < a.text[E]) ? 2 : 1;
        }
        if (lane == 0) {
            a.dec[g] = dec;
            // a group of this difference whose run ends inside it stays tied: out of the histogram, or the next iteration
            // of the attempt picks the same difference again and repeats the passes over the text for nothing
            if (!dec && !a.bad[g] && (u64)a.gd[g] == a.d) a.bad[g] = 1;
        }
    }
}
// one thread per list position: the members of a decided group go to the group's SA slots in ascending or descending order
// of position, every one a singleton; the group is retired (bad = 1) for the next difference's histogram
__global__ __launch_bounds__(256) void per_apply_kernel(PerArgs a) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < a.M; j += stride) {
        const u32 g = a.gid[j];
        const u8 dec = a.dec[g];
        if (!dec) continue;
        const u32 gs = a.gstart[g], ge = a.gstart[g + 1];
        const u32 r = (u32)j - gs, m = ge - gs;
        const u32 slot = a.apos[j];
        const u32 v = a.aidx[gs + (dec == 2 ? (m - 1 - r) : r)];
        a.sa[slot] = v;
        a.gflags[slot] = 1;
        a.done[j] = 1;
        if (a.isa) a.isa[v] = slot;
        if (r == 0) a.bad[g] = 1;
    }
}

}  // namespace sa
