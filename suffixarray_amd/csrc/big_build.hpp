// big_build.hpp -- suffix arrays of texts of MORE than 2^32 - 2 bytes: 64-bit suffix indices throughout.
//
// The reference's libsais64 falls through to a true 64-bit construction for n > INT32_MAX (libsais64.c:6684 ->
// libsais64_main, 6480-6519); the product pipeline of sa_build.hpp keeps unsigned 32-bit suffix indices and covers
// n <= 2^32 - 2.  This file is the path beyond: sa_hip_libsais64 answers for every n the GPU's memory holds instead of
// refusing.  It is a functional completion, not the benchmark path -- plain three-kernel radix passes (per-tile counts,
// scan, stable scatter through LDS) over (u64 key, u64 suffix) records and Larsson-Sadakane prefix doubling with
// discarding on what the initial sort leaves tied:
//   1. byte histogram -> alphabet compaction (b bits per character), key = the first k = floor(64 / b) characters;
//   2. LSD radix sort of (key, suffix) over the key's bits: the suffixes ordered by their first k characters;
//   3. group heads, ISA[suffix] = SA slot of its group's head, the tied records compacted into lists
//      (SA slot, suffix, dense group id);
//   4. rounds, h = k, 2k, 4k, ...: key2 = ISA[suffix + h] + 1 (0 past the end: a suffix that ends sorts first);
//      stable sort of the list by key2, then by group id (two pair sorts: group id and rank together need 65 bits at
//      n < 2^33); SA slots of the list rewritten, new heads where (group, key2) changes, ISA of the list updated,
//      singletons dropped.
// Memory: 8 n (caller's SA, used as one of the sort's buffers) + 16 n keys + 8 n second suffix buffer + 8 n ISA + n text
// = 41 n bytes during the initial sort (180 GB at n = 4.4e9), 17 n + 90 bytes per tied record afterwards.
#pragma once
#include "sa_build.hpp"

namespace sa {
namespace big {

constexpr int BG_BLOCK = 512;
constexpr int BG_ITEMS = 16;
constexpr u32 BG_TILE = BG_BLOCK * BG_ITEMS;   // 8192 records per workgroup
constexpr int BG_WAVES = BG_BLOCK / WAVE;

// ---- pair sort: one 8-bit digit per pass -------------------------------------------------------------------------------
// per-tile digit counts, digit-major (th[d * ntiles + tile]): an exclusive scan of the array in that order is every
// tile's destination per digit
__global__ __launch_bounds__(BG_BLOCK) void bg_hist_kernel(const u64* __restrict__ keys, u64 cnt, int shift, u32 ntiles, u32* __restrict__ th) {
    constexpr int CS = 257;
    __shared__ u32 s_h[8 * CS];
    for (int i = threadIdx.x; i < 8 * CS; i += BG_BLOCK) s_h[i] = 0;
    __syncthreads();
    u32* my = s_h + (threadIdx.x & 7) * CS;
    const u64 base = (u64)blockIdx.x * BG_TILE;
#pragma unroll 4
    for (int it = 0; it < BG_ITEMS; ++it) {
        const u64 j = base + (u64)it * BG_BLOCK + threadIdx.x;
        if (j < cnt) atomicAdd(&my[(u32)(keys[j] >> shift) & 255u], 1u);
    }
    sync_lds();
    if (threadIdx.x < 256) {
        u32 c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c += s_h[k * CS + threadIdx.x];
        th[(u64)threadIdx.x * ntiles + blockIdx.x] = c;
    }
}

// exclusive scan u32 -> u64 in three steps (reduce per 8192 elements, scan of the partial sums by one workgroup, apply)
constexpr u32 SC_BLOCK = 1024, SC_ITEMS = 8, SC_TILE = SC_BLOCK * SC_ITEMS;
__device__ __forceinline__ u64 bg_block_excl_scan(u64 v, u64* s_w, u64* total) {   // SC_BLOCK threads; returns the exclusive prefix of v
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    u64 off = 0, tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        const u64 t = s_w[w];
        if (w < wave) off += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return off + incl - v;
}
__global__ __launch_bounds__(SC_BLOCK) void bg_scan_reduce_kernel(const u32* __restrict__ in, u64 len, u64* __restrict__ part) {
    __shared__ u64 s_w[SC_BLOCK / 64];
    const u64 base = (u64)blockIdx.x * SC_TILE + (u64)threadIdx.x * SC_ITEMS;
    u64 s = 0;
#pragma unroll
    for (u32 e = 0; e < SC_ITEMS; ++e) if (base + e < len) s += in[base + e];
    u64 tot;
    (void)bg_block_excl_scan(s, s_w, &tot);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(SC_BLOCK) void bg_scan_parts_kernel(u64* __restrict__ part, u64 nparts) {   // in place, exclusive; part[nparts] = total
    __shared__ u64 s_w[SC_BLOCK / 64];
    u64 carry = 0;
    for (u64 base = 0; base < nparts; base += SC_BLOCK) {
        const u64 i = base + threadIdx.x;
        const u64 v = i < nparts ? part[i] : 0;
        u64 tot;
        const u64 ex = bg_block_excl_scan(v, s_w, &tot);
        if (i < nparts) part[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) part[nparts] = carry;
}
__global__ __launch_bounds__(SC_BLOCK) void bg_scan_apply_kernel(const u32* __restrict__ in, u64 len, const u64* __restrict__ part, u64* __restrict__ out) {
    __shared__ u64 s_w[SC_BLOCK / 64];
    const u64 base = (u64)blockIdx.x * SC_TILE + (u64)threadIdx.x * SC_ITEMS;
    u32 v[SC_ITEMS];
    u64 s = 0;
#pragma unroll
    for (u32 e = 0; e < SC_ITEMS; ++e) { v[e] = (base + e < len) ? in[base + e] : 0u; s += v[e]; }
    u64 tot;
    u64 run = bg_block_excl_scan(s, s_w, &tot) + part[blockIdx.x];
#pragma unroll
    for (u32 e = 0; e < SC_ITEMS; ++e) { if (base + e < len) out[base + e] = run; run += v[e]; }
}

// stable scatter of one tile: ballot-match ranking per wave (wave_rank, radix_sort.hpp), the records reordered in LDS so that
// every digit's run leaves as contiguous stores; keys first, then the values through the same LDS
template <bool FULL>
__device__ __forceinline__ void bg_scatter_tile(const u64* __restrict__ kin, const u64* __restrict__ vin, u64* __restrict__ kout, u64* __restrict__ vout,
                                                const u64 base, const u32 tile_n, const int shift, const u32 ntiles, const u64* __restrict__ off,
                                                u64* s_rec, u32* s_wh, u64* s_gd, u32* s_wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 woff = (u32)wave * (WAVE * BG_ITEMS) + lane;
    u64 key[BG_ITEMS];
#pragma unroll
    for (int j = 0; j < BG_ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        key[j] = (FULL || p < tile_n) ? kin[base + p] : ~0ull;
    }
    u32 rd[BG_ITEMS];
    u32* wh = s_wh + wave * RADIX;
    wave_rank<FULL, u64, BG_ITEMS>(key, shift, 255u, woff, tile_n, wh, rd);
    __syncthreads();
    u32 excl = 0, c = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < BG_WAVES; ++w) {
            const u32 t = s_wh[w * RADIX + tid];
            s_wh[w * RADIX + tid] = c;
            c += t;
        }
        u32 incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wsum[wave] = incl;
        excl = incl - c;
    }
    __syncthreads();
    if (tid < RADIX) {
        for (int i = 0; i < wave; ++i) excl += s_wsum[i];
#pragma unroll
        for (int w = 0; w < BG_WAVES; ++w) s_wh[w * RADIX + tid] += excl;
        s_gd[tid] = off[(u64)tid * ntiles + blockIdx.x] - excl;
    }
    __syncthreads();
    u32 pos[BG_ITEMS];
#pragma unroll
    for (int j = 0; j < BG_ITEMS; ++j) {
        pos[j] = wh[rd[j] >> 16] + (rd[j] & 0xFFFFu);
        if (FULL || (woff + j * WAVE) < tile_n) s_rec[pos[j]] = key[j];
    }
    __syncthreads();
    u64 g[BG_ITEMS];
#pragma unroll
    for (int k = 0; k < BG_ITEMS; ++k) {
        const u32 p = k * BG_BLOCK + tid;
        g[k] = 0;
        if (FULL || p < tile_n) {
            const u64 kk = s_rec[p];
            g[k] = s_gd[(u32)(kk >> shift) & 255u] + p;
            kout[g[k]] = kk;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BG_ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        if (FULL || p < tile_n) s_rec[pos[j]] = vin[base + p];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BG_ITEMS; ++k) {
        const u32 p = k * BG_BLOCK + tid;
        if (FULL || p < tile_n) vout[g[k]] = s_rec[p];
    }
}
__global__ __launch_bounds__(BG_BLOCK, 4) void bg_scatter_kernel(const u64* __restrict__ kin, const u64* __restrict__ vin, u64* __restrict__ kout,
                                                                 u64* __restrict__ vout, u64 cnt, int shift, u32 ntiles, const u64* __restrict__ off) {
    __shared__ u64 s_rec[BG_TILE];
    __shared__ u32 s_wh[BG_WAVES * RADIX];
    __shared__ u64 s_gd[RADIX];
    __shared__ u32 s_wsum[RADIX / WAVE];
    for (int i = threadIdx.x; i < BG_WAVES * RADIX; i += BG_BLOCK) s_wh[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * BG_TILE;
    const u64 rest = cnt - base;
    if (rest >= BG_TILE) bg_scatter_tile<true>(kin, vin, kout, vout, base, BG_TILE, shift, ntiles, off, s_rec, s_wh, s_gd, s_wsum);
    else bg_scatter_tile<false>(kin, vin, kout, vout, base, (u32)rest, shift, ntiles, off, s_rec, s_wh, s_gd, s_wsum);
}

// ---- initial keys ----------------------------------------------------------------------------------------------------------
// key[p] = the first k characters of suffix p as b-bit codes, MSB first, right aligned (0 past the end); sixteen consecutive
// positions per thread through a rolling window
__global__ __launch_bounds__(256) void bg_keygen_kernel(const u8* __restrict__ text, u64 n, CodeMap map, int b, int k, u64* __restrict__ keys,
                                                        u64* __restrict__ idx) {
    __shared__ u16 s_map[256];
    s_map[threadIdx.x] = map.code[threadIdx.x];
    __syncthreads();
    const u64 p0 = ((u64)blockIdx.x * 256 + threadIdx.x) * 16;
    if (p0 >= n) return;
    const int kb = k * b;
    const u64 mask = kb >= 64 ? ~0ull : ((1ull << kb) - 1ull);
    u64 win = 0;
    for (int c = 0; c + 1 < k; ++c) {
        const u64 x = p0 + (u64)c;
        win = (win << b) | (x < n ? (u64)s_map[text[x]] : 0ull);
    }
    for (int i = 0; i < 16; ++i) {
        const u64 p = p0 + (u64)i, x = p + (u64)(k - 1);
        win = ((win << b) | (x < n ? (u64)s_map[text[x]] : 0ull)) & mask;
        if (p < n) { keys[p] = win; idx[p] = p; }
    }
}

// ---- group heads, ranks, the tied records as lists -----------------------------------------------------------------------
// A list position q (all of the suffix array after the initial sort: slot == nullptr, the position IS the SA slot; later the
// tied records only) heads a group when its key differs from its predecessor's: kA, and kB where given (group id, key2).
// Three passes: per-tile aggregates, their exclusive scan (one workgroup), apply:
//   isa[idx[q]] = SA slot of the head of q's group;  a record whose group has more than one member goes to the next lists
//   (its SA slot, its suffix, the dense id of its group among the tied groups).
struct FlagAgg {
    u64 lasthead;   // 1 + the last position that heads a group, 0: none
    u64 nact;       // tied records
    u64 nhead;      // tied groups
};
struct FlagArgs {
    const u64* kA;
    const u64* kB;      // or nullptr
    const u64* slot;    // or nullptr: the position is the SA slot
    const u64* idx;
    u64 m;
    u64* isa;
    u64* out_slot;
    u64* out_idx;
    u64* out_gid;
};
__device__ __forceinline__ bool bg_is_head(const FlagArgs& a, u64 q) {
    if (q == 0) return true;
    if (a.kA[q] != a.kA[q - 1]) return true;
    return a.kB && a.kB[q] != a.kB[q - 1];
}
__device__ __forceinline__ FlagAgg bg_combine(const FlagAgg& x, const FlagAgg& y) {   // x before y
    FlagAgg r;
    r.lasthead = y.lasthead ? y.lasthead : x.lasthead;
    r.nact = x.nact + y.nact;
    r.nhead = x.nhead + y.nhead;
    return r;
}
// exclusive scan of one aggregate per thread over the workgroup (BG_BLOCK threads); *total = the workgroup's aggregate
__device__ __forceinline__ FlagAgg bg_block_scan_agg(FlagAgg v, FlagAgg* s_w, FlagAgg* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    FlagAgg incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        FlagAgg t;
        t.lasthead = __shfl_up(incl.lasthead, o); t.nact = __shfl_up(incl.nact, o); t.nhead = __shfl_up(incl.nhead, o);
        if (lane >= o) incl = bg_combine(t, incl);
    }
    FlagAgg excl;   // exclusive inside the wave
    excl.lasthead = __shfl_up(incl.lasthead, 1); excl.nact = __shfl_up(incl.nact, 1); excl.nhead = __shfl_up(incl.nhead, 1);
    if (lane == 0) { excl.lasthead = 0; excl.nact = 0; excl.nhead = 0; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    FlagAgg off{0, 0, 0}, tot{0, 0, 0};
    for (int w = 0; w < BG_WAVES; ++w) {
        const FlagAgg t = s_w[w];
        if (w < wave) off = bg_combine(off, t);
        tot = bg_combine(tot, t);
    }
    __syncthreads();
    *total = tot;
    return bg_combine(off, excl);
}
// a thread looks at BG_ITEMS CONSECUTIVE positions (head / tied bits of them in two masks)
__device__ __forceinline__ FlagAgg bg_thread_flags(const FlagArgs& a, u64 q0, u32& hmask, u32& amask) {
    FlagAgg g{0, 0, 0};
    hmask = 0; amask = 0;
    if (q0 >= a.m) return g;
    bool h = bg_is_head(a, q0);
#pragma unroll
    for (int e = 0; e < BG_ITEMS; ++e) {
        const u64 q = q0 + e;
        if (q < a.m) {
            const bool hn = (q + 1 == a.m) || bg_is_head(a, q + 1);
            const bool act = !(h && hn);
            if (h) { hmask |= 1u << e; g.lasthead = q + 1; }
            if (act) { amask |= 1u << e; ++g.nact; if (h) ++g.nhead; }
            h = hn;
        }
    }
    return g;
}
__global__ __launch_bounds__(BG_BLOCK) void bg_flags_reduce_kernel(FlagArgs a, FlagAgg* __restrict__ part) {
    __shared__ FlagAgg s_w[BG_WAVES];
    u32 hm, am;
    const FlagAgg g = bg_thread_flags(a, (u64)blockIdx.x * BG_TILE + (u64)threadIdx.x * BG_ITEMS, hm, am);
    FlagAgg tot;
    (void)bg_block_scan_agg(g, s_w, &tot);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(BG_BLOCK) void bg_flags_scan_kernel(FlagAgg* __restrict__ part, u64 nparts) {   // in place, exclusive; part[nparts] = total
    __shared__ FlagAgg s_w[BG_WAVES];
    FlagAgg carry{0, 0, 0};
    for (u64 base = 0; base < nparts; base += BG_BLOCK) {
        const u64 i = base + threadIdx.x;
        FlagAgg v{0, 0, 0};
        if (i < nparts) v = part[i];
        FlagAgg tot;
        const FlagAgg ex = bg_block_scan_agg(v, s_w, &tot);
        if (i < nparts) part[i] = bg_combine(carry, ex);
        carry = bg_combine(carry, tot);
    }
    if (threadIdx.x == 0) part[nparts] = carry;
}
__global__ __launch_bounds__(BG_BLOCK) void bg_flags_apply_kernel(FlagArgs a, const FlagAgg* __restrict__ part) {
    __shared__ FlagAgg s_w[BG_WAVES];
    const u64 q0 = (u64)blockIdx.x * BG_TILE + (u64)threadIdx.x * BG_ITEMS;
    u32 hm, am;
    const FlagAgg g = bg_thread_flags(a, q0, hm, am);
    FlagAgg tot;
    FlagAgg run = bg_combine(part[blockIdx.x], bg_block_scan_agg(g, s_w, &tot));   // everything before q0
    if (q0 >= a.m) return;
#pragma unroll
    for (int e = 0; e < BG_ITEMS; ++e) {
        const u64 q = q0 + e;
        if (q < a.m) {
            const bool h = (hm >> e) & 1u, act = (am >> e) & 1u;
            if (h) { run.lasthead = q + 1; if (act) ++run.nhead; }
            const u64 hq = run.lasthead - 1;            // q == 0 heads a group: never 0 - 1
            const u64 sfx = a.idx[q];
            a.isa[sfx] = a.slot ? a.slot[hq] : hq;
            if (act) {
                a.out_slot[run.nact] = a.slot ? a.slot[q] : q;
                a.out_idx[run.nact] = sfx;
                a.out_gid[run.nact] = run.nhead - 1;
                ++run.nact;
            }
        }
    }
}

// ---- the rounds ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bg_key2_kernel(const u64* __restrict__ idx, u64 m, u64 n, u64 h, const u64* __restrict__ isa, u64* __restrict__ key2,
                                                      u64* __restrict__ perm) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += stride) {
        const u64 x = idx[p] + h;
        key2[p] = x < n ? isa[x] + 1 : 0ull;
        perm[p] = p;
    }
}
__global__ __launch_bounds__(256) void bg_gather_kernel(const u64* __restrict__ src, const u64* __restrict__ perm, u64 m, u64* __restrict__ dst) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += stride) dst[q] = src[perm[q]];
}
// the list in its new order: suffixes and key2 by the sort's permutation, the SA slots of the list rewritten
__global__ __launch_bounds__(256) void bg_permute_kernel(const u64* __restrict__ idx, const u64* __restrict__ key2, const u64* __restrict__ perm,
                                                         const u64* __restrict__ slot, u64 m, u64* __restrict__ nidx, u64* __restrict__ nk2,
                                                         u64* __restrict__ sa) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += stride) {
        const u64 p = perm[q];
        const u64 s = idx[p];
        nidx[q] = s;
        nk2[q] = key2[p];
        sa[slot[q]] = s;
    }
}

// ---- sufcheck with 64-bit indices (tests): SA is a permutation of [0, n) and neighbours are in order ---------------------
__global__ __launch_bounds__(256) void bg_inverse_kernel(const u64* __restrict__ sa, u64 n, u64* __restrict__ inv, unsigned long long* __restrict__ bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const u64 s = sa[j];
        if (s < n) inv[s] = j; else atomicAdd(bad, 1ull);
    }
}
__global__ __launch_bounds__(256) void bg_order_kernel(const u8* __restrict__ text, const u64* __restrict__ sa, u64 n, const u64* __restrict__ inv,
                                                       unsigned long long* __restrict__ bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const u64 b = sa[j];
        if (b >= n) continue;   // counted above
        bool ok = inv[b] == j;  // (two slots with the same suffix: one of them fails here)
        if (ok && j > 0) {
            const u64 a = sa[j - 1];
            if (a >= n) continue;
            const u8 ta = text[a], tb = text[b];
            if (ta != tb) ok = ta < tb;
            else {
                // equal first characters: the order of the suffixes one further on decides; the one that ends there sorts first
                const bool ea = a + 1 >= n, eb = b + 1 >= n;
                ok = ea ? !eb : (!eb && inv[a + 1] < inv[b + 1]);
            }
        }
        if (!ok) atomicAdd(bad, 1ull);
    }
}

struct BigStats {
    u32 sigma = 0, bits_per_symbol = 0, initial_chars = 0, sort_passes = 0, rounds = 0;
    u64 tied_after_sort = 0, tied_total = 0;
    float total_ms = 0.f;
};

struct BigBuilder {
    hipStream_t stream = nullptr;
    DevBuf keysA, keysB, idxB, isa, th, off, part, fpart, small;
    DevBuf l_slot[2], l_idx[2], l_gid[2], r_key2, r_perm, r_k1, r_p1, r_gk, r_nk2;
    BigStats stats;

    void destroy() {
        DevBuf* all[] = {&keysA, &keysB, &idxB, &isa, &th, &off, &part, &fpart, &small, &l_slot[0], &l_slot[1], &l_idx[0], &l_idx[1], &l_gid[0],
                         &l_gid[1], &r_key2, &r_perm, &r_k1, &r_p1, &r_gk, &r_nk2};
        for (DevBuf* b : all) b->release();
    }
    ~BigBuilder() { destroy(); }

    // stable LSD sort of cnt (key, value) pairs over key bits [0, bits): result pointers in *kres / *vres (one of the two pairs of buffers)
    int sort_pairs(u64* k0, u64* v0, u64* k1, u64* v1, u64 cnt, int bits, u64** kres, u64** vres) {
        *kres = k0; *vres = v0;
        if (cnt < 2 || bits <= 0) return 0;
        const u64 nt64 = (cnt + BG_TILE - 1) / BG_TILE;
        if (nt64 > 0x7FFFFFFFull) return fail(SA_HIP_EINVAL, "big sort: too many tiles");
        const u32 ntiles = (u32)nt64;
        const u64 len = (u64)RADIX * ntiles;
        const u64 nparts = (len + SC_TILE - 1) / SC_TILE;
        int rc;
        if ((rc = th.ensure(len * 4 + 64))) return rc;
        if ((rc = off.ensure(len * 8 + 64))) return rc;
        if ((rc = part.ensure((nparts + 1) * 8 + 64))) return rc;
        u64 *ki = k0, *vi = v0, *ko = k1, *vo = v1;
        for (int shift = 0; shift < bits; shift += 8) {
            hipLaunchKernelGGL(bg_hist_kernel, dim3(ntiles), dim3(BG_BLOCK), 0, stream, (const u64*)ki, cnt, shift, ntiles, th.as<u32>());
            hipLaunchKernelGGL(bg_scan_reduce_kernel, dim3((u32)nparts), dim3(SC_BLOCK), 0, stream, (const u32*)th.as<u32>(), len, part.as<u64>());
            hipLaunchKernelGGL(bg_scan_parts_kernel, dim3(1), dim3(SC_BLOCK), 0, stream, part.as<u64>(), nparts);
            hipLaunchKernelGGL(bg_scan_apply_kernel, dim3((u32)nparts), dim3(SC_BLOCK), 0, stream, (const u32*)th.as<u32>(), len, (const u64*)part.as<u64>(),
                               off.as<u64>());
            hipLaunchKernelGGL(bg_scatter_kernel, dim3(ntiles), dim3(BG_BLOCK), 0, stream, (const u64*)ki, (const u64*)vi, ko, vo, cnt, shift, ntiles,
                               (const u64*)off.as<u64>());
            u64* t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
            ++stats.sort_passes;
        }
        SA_HIP_CHECK(hipGetLastError());
        *kres = ki; *vres = vi;
        return 0;
    }

    // heads / ranks / next lists of a list of m records; *m_out, *g_out = tied records and groups that remain
    int flags_pass(const u64* kA, const u64* kB, const u64* slot, const u64* idx, u64 m, int out, u64* m_out, u64* g_out) {
        const u64 nt64 = (m + BG_TILE - 1) / BG_TILE;
        if (nt64 > 0x7FFFFFFFull) return fail(SA_HIP_EINVAL, "big flags: too many tiles");
        const u32 ntiles = (u32)nt64;
        int rc;
        if ((rc = fpart.ensure(((size_t)ntiles + 1) * sizeof(FlagAgg) + 64))) return rc;
        FlagArgs a{};
        a.kA = kA; a.kB = kB; a.slot = slot; a.idx = idx; a.m = m; a.isa = isa.as<u64>();
        hipLaunchKernelGGL(bg_flags_reduce_kernel, dim3(ntiles), dim3(BG_BLOCK), 0, stream, a, fpart.as<FlagAgg>());
        hipLaunchKernelGGL(bg_flags_scan_kernel, dim3(1), dim3(BG_BLOCK), 0, stream, fpart.as<FlagAgg>(), (u64)ntiles);
        FlagAgg tot;
        SA_HIP_CHECK(hipMemcpyAsync(&tot, fpart.as<FlagAgg>() + ntiles, sizeof tot, hipMemcpyDeviceToHost, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        *m_out = tot.nact; *g_out = tot.nhead;
        if ((rc = l_slot[out].ensure((size_t)tot.nact * 8 + 64))) return rc;
        if ((rc = l_idx[out].ensure((size_t)tot.nact * 8 + 64))) return rc;
        if ((rc = l_gid[out].ensure((size_t)tot.nact * 8 + 64))) return rc;
        a.out_slot = l_slot[out].as<u64>(); a.out_idx = l_idx[out].as<u64>(); a.out_gid = l_gid[out].as<u64>();
        hipLaunchKernelGGL(bg_flags_apply_kernel, dim3(ntiles), dim3(BG_BLOCK), 0, stream, a, (const FlagAgg*)fpart.as<FlagAgg>());
        SA_HIP_CHECK(hipGetLastError());
        return 0;
    }

    // text_dev: n bytes (16-byte aligned); sa_out: n entries of 8 bytes on the device (the libsais64 layout: the suffix indices are
    // below 2^63, u64 and int64 are the same bits)
    int build(const u8* text_dev, u64 n, u64* sa_out) {
        stats = BigStats{};
        if (n == 0) return 0;
        int rc;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        SA_HIP_CHECK(hipEventCreate(&e0));
        SA_HIP_CHECK(hipEventCreate(&e1));
        struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } guard{e0, e1};
        // the large buffers first (allocating 4 x 8 n bytes takes the driver longer than the build: outside the timed region)
        if ((rc = small.ensure(4096))) return rc;
        if ((rc = keysA.ensure(n * 8 + 64))) return rc;
        if ((rc = keysB.ensure(n * 8 + 64))) return rc;
        if ((rc = idxB.ensure(n * 8 + 64))) return rc;
        if ((rc = isa.ensure(n * 8 + 64))) return rc;
        {
            const u64 ntiles = (n + BG_TILE - 1) / BG_TILE, len = (u64)RADIX * ntiles;
            if ((rc = th.ensure(len * 4 + 64)) || (rc = off.ensure(len * 8 + 64)) || (rc = part.ensure(((len + SC_TILE - 1) / SC_TILE + 1) * 8 + 64)) ||
                (rc = fpart.ensure((ntiles + 1) * sizeof(FlagAgg) + 64))) return rc;
        }
        SA_HIP_CHECK(hipEventRecord(e0, stream));
        // 1. alphabet
        u64* dh = small.as<u64>();
        SA_HIP_CHECK(hipMemsetAsync(dh, 0, 256 * sizeof(u64), stream));
        hipLaunchKernelGGL(byte_hist_kernel, dim3(stream_grid(n, 256 * 64)), dim3(256), 0, stream, text_dev, n, dh);
        u64 freq[256];
        SA_HIP_CHECK(hipMemcpyAsync(freq, dh, sizeof freq, hipMemcpyDeviceToHost, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        CodeMap map;
        memset(&map, 0, sizeof map);
        u32 sigma = 0;
        for (int c = 0; c < 256; ++c) if (freq[c]) map.code[c] = (u16)(++sigma);
        int b = bits_for((u64)sigma + 1);
        if (b == 0) b = 1;
        int k = 64 / b;
        if ((u64)k > n) k = (int)n;   // (a key longer than the text adds nothing)
        if (k < 1) k = 1;
        stats.sigma = sigma; stats.bits_per_symbol = (u32)b; stats.initial_chars = (u32)k;
        // 2. keys + initial sort; the caller's array is one of the two suffix buffers
        hipLaunchKernelGGL(bg_keygen_kernel, dim3((u32)((n + 4095) / 4096)), dim3(256), 0, stream, text_dev, n, map, b, k, keysA.as<u64>(), sa_out);
        u64 *kres, *vres;
        if ((rc = sort_pairs(keysA.as<u64>(), sa_out, keysB.as<u64>(), idxB.as<u64>(), n, k * b, &kres, &vres))) return rc;
        if (vres != sa_out) SA_HIP_CHECK(hipMemcpyAsync(sa_out, vres, n * 8, hipMemcpyDeviceToDevice, stream));
        // 3. heads, ranks, the tied records
        u64 M = 0, G = 0;
        int cur = 0;
        if ((rc = flags_pass(kres, nullptr, nullptr, sa_out, n, cur, &M, &G))) return rc;
        stats.tied_after_sort = M;
        keysA.release(); keysB.release(); idxB.release(); th.release(); off.release();   // the rounds allocate by the size of the tied set
        // 4. prefix doubling on the lists
        const int rbits = bits_for(n + 2);
        for (u64 h = (u64)k; M > 0; h *= 2) {
            stats.tied_total += M;
            ++stats.rounds;
            if ((rc = r_key2.ensure(M * 8 + 64)) || (rc = r_perm.ensure(M * 8 + 64)) || (rc = r_k1.ensure(M * 8 + 64)) || (rc = r_p1.ensure(M * 8 + 64)) ||
                (rc = r_gk.ensure(M * 8 + 64)) || (rc = r_nk2.ensure(M * 8 + 64))) return rc;
            const u32 grid = stream_grid(M, 1024);
            hipLaunchKernelGGL(bg_key2_kernel, dim3(grid), dim3(256), 0, stream, (const u64*)l_idx[cur].as<u64>(), M, n, h, (const u64*)isa.as<u64>(),
                               r_key2.as<u64>(), r_perm.as<u64>());
            // by key2 (the sort moves copies: r_key2 stays in list order for the permutation below)
            SA_HIP_CHECK(hipMemcpyAsync(r_gk.p, r_key2.p, M * 8, hipMemcpyDeviceToDevice, stream));
            u64 *k1, *p1;
            if ((rc = sort_pairs(r_gk.as<u64>(), r_perm.as<u64>(), r_k1.as<u64>(), r_p1.as<u64>(), M, rbits, &k1, &p1))) return rc;
            // then, stable, by group id
            u64* gk = (k1 == r_gk.as<u64>()) ? r_k1.as<u64>() : r_gk.as<u64>();         // the key buffer the result is NOT in
            u64* pfree = (p1 == r_perm.as<u64>()) ? r_p1.as<u64>() : r_perm.as<u64>();
            hipLaunchKernelGGL(bg_gather_kernel, dim3(grid), dim3(256), 0, stream, (const u64*)l_gid[cur].as<u64>(), (const u64*)p1, M, gk);
            u64 *k2, *p2;
            if ((rc = sort_pairs(gk, p1, k1, pfree, M, bits_for(G + 1), &k2, &p2))) return rc;
            // the list in its new order (the group ids of a list are non-decreasing and the sort is stable: position q keeps its group)
            const int nxt = cur ^ 1;
            if ((rc = l_idx[nxt].ensure(M * 8 + 64))) return rc;
            hipLaunchKernelGGL(bg_permute_kernel, dim3(grid), dim3(256), 0, stream, (const u64*)l_idx[cur].as<u64>(), (const u64*)r_key2.as<u64>(),
                               (const u64*)p2, (const u64*)l_slot[cur].as<u64>(), M, l_idx[nxt].as<u64>(), r_nk2.as<u64>(), sa_out);
            // new heads, ranks of the list's suffixes, what is still tied (the new lists are written to the other set of buffers;
            // the permuted suffixes sit in l_idx[nxt], which the pass also writes: through a copy)
            SA_HIP_CHECK(hipMemcpyAsync(r_perm.p, l_idx[nxt].p, M * 8, hipMemcpyDeviceToDevice, stream));
            u64 M2 = 0, G2 = 0;
            if ((rc = flags_pass(l_gid[cur].as<u64>(), r_nk2.as<u64>(), l_slot[cur].as<u64>(), r_perm.as<u64>(), M, nxt, &M2, &G2))) return rc;
            M = M2; G = G2; cur = nxt;
            if (h > n) break;   // (cannot happen: at h >= n every key2 is 0 or final)
        }
        SA_HIP_CHECK(hipEventRecord(e1, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        (void)hipEventElapsedTime(&stats.total_ms, e0, e1);
        if (M) return fail(SA_HIP_EHIP, "big build: rounds ended with tied suffixes");
        return 0;
    }

    // sufcheck: number of slots that violate "permutation in suffix order" (0 = the array is THE suffix array)
    int verify(const u8* text_dev, const u64* sa, u64 n, u64* violations) {
        *violations = 0;
        if (n == 0) return 0;
        int rc;
        if ((rc = isa.ensure(n * 8 + 64))) return rc;
        if ((rc = small.ensure(4096))) return rc;
        unsigned long long* bad = small.as<unsigned long long>();
        SA_HIP_CHECK(hipMemsetAsync(bad, 0, 8, stream));
        SA_HIP_CHECK(hipMemsetAsync(isa.p, 0xFF, n * 8, stream));
        const u32 grid = stream_grid(n, 1024);
        hipLaunchKernelGGL(bg_inverse_kernel, dim3(grid), dim3(256), 0, stream, sa, n, isa.as<u64>(), bad);
        hipLaunchKernelGGL(bg_order_kernel, dim3(grid), dim3(256), 0, stream, text_dev, sa, n, (const u64*)isa.as<u64>(), bad);
        unsigned long long v = 0;
        SA_HIP_CHECK(hipMemcpyAsync(&v, bad, 8, hipMemcpyDeviceToHost, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        *violations = v;
        return 0;
    }
};

}  // namespace big
}  // namespace sa
