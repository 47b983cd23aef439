// round_sort.hpp -- sort of the records of one refinement round, group by group, inside LDS.
//
// The records of a round arrive ordered by their group id (the active list is in SA order) and their keys are
// gid << gid_shift | (next characters, or the rank of the suffix h further on): the global 8-pass LSD sort of
// sa_build.hpp re-sorts the group ids it was given in order.  A round is really a SEGMENTED sort: every group is
// sorted by the low key bits and stays where it is.  So:
//   * the list is cut into tiles of whole groups: tile t = the groups that START in [t*T, (t+1)*T)
//     (a_t = start of the first group that starts at or after t*T, O(1) from gstart[gid]);
//   * a tile of at most C = 4096 records is sorted by ONE workgroup entirely in LDS: keys and values are read
//     once and written once (24 instead of 8 * 24 bytes of HBM traffic per record), and the group id shrinks to
//     the 12 bits that tell the groups of one tile apart (local gid = gid - gid of the tile's first record):
//     ceil((low bits + 12) / 8) passes -- 6 for a chunk round of 7 five-bit characters -- of ballot-match
//     ranking (wave_rank, stable) with the records held in registers between passes;
//   * a tile that would exceed C ends with a group larger than C - T: that group alone goes to a compact
//     list which the global sort handles (radix_sort_pairs) and is copied back into place.
// Output: the sorted records at the positions their groups occupy in the list, exactly what the global sort
// of the whole list would have produced.
#pragma once
#include "radix_sort.hpp"

namespace sa {

// 256 threads x 16 records.  512 threads (tiles of 8192, 13-bit local ids, 72 KB of LDS: two workgroups per CU, fewer
// groups through the big-group list) measured SLOWER on the same box: names 1e8 12.9 against 12.0 ms, words 14.9 against 13.95.
#ifndef SA_LOC_BLOCK
#define SA_LOC_BLOCK 256
#endif
constexpr int LOC_BLOCK = SA_LOC_BLOCK;          // 256 or 512 threads
constexpr int LOC_ITEMS = 16;
constexpr u32 LOC_CAP = LOC_BLOCK * LOC_ITEMS;   // 4096 (8192) records per workgroup
// nominal tile = 7/8 of the capacity: groups of up to LOC_CAP / 8 + 1 records never overflow; slack 1/2, 1/4, 1/8, 1/16, 1/32 of
// the capacity measured (names 1e8): 12.46, 12.04, 11.74-11.92, 12.07, 12.20 ms -- padding lanes against overflowing groups
#ifndef SA_LOC_SLACK_DIV
#define SA_LOC_SLACK_DIV 8
#endif
constexpr u32 LOC_TILE = LOC_CAP - LOC_CAP / SA_LOC_SLACK_DIV;
constexpr int LOC_GID_BITS = (LOC_BLOCK == 512) ? 13 : 12;   // local group ids < LOC_CAP

// gstart[g] = first record of group g (gid is dense and ascending); gstart[G] = M
__global__ __launch_bounds__(256) void group_starts_kernel(const u32* __restrict__ gid, u32 m_count, u32 groups,
                                                           u32* __restrict__ gstart) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x; m < m_count; m += stride) {
        const u32 g = gid[m];
        if (m == 0 || gid[m - 1] != g) gstart[g] = (u32)m;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) gstart[groups] = m_count;
}

struct LocTile {
    u32 begin;       // a_t
    u32 local_end;   // records [begin, local_end) are sorted in LDS
    u32 end;         // a_{t+1}; records [local_end, end) = one big group (or none)
    u32 big_off;     // exclusive prefix of (end - local_end) over the tiles, filled by loc_scan_kernel
};

template <u32 TILE>
__device__ __forceinline__ u32 loc_tile_start(const u32* gid, const u32* gstart, u32 m_count, u64 t) {
    const u64 p = t * TILE;
    if (p >= m_count) return m_count;
    const u32 g = gid[p];
    return (gstart[g] == (u32)p) ? (u32)p : gstart[g + 1];   // p inside a group: that group belongs to the tile before
}

// TILE = nominal tile length, CAP = what a workgroup can hold (the tile-local round sort: LOC_TILE, LOC_CAP; the group
// finisher has its own pair, group_finish.hpp)
template <u32 TILE = LOC_TILE, u32 CAP = LOC_CAP>
__global__ __launch_bounds__(256) void loc_plan_kernel(const u32* __restrict__ gid, const u32* __restrict__ gstart, u32 m_count,
                                                       u32 ntiles, LocTile* __restrict__ tiles) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const u32 a = loc_tile_start<TILE>(gid, gstart, m_count, t);
    const u32 b = loc_tile_start<TILE>(gid, gstart, m_count, (u64)t + 1);
    LocTile lt;
    lt.begin = a;
    lt.end = b < a ? a : b;
    lt.local_end = lt.end;
    if (lt.end - a > CAP) {
        // the group that straddles (t+1)*T makes the tile too long: everything before it fits (it starts before
        // (t+1)*T, so [a, its start) has fewer than T records)
        lt.local_end = gstart[gid[(u64)(t + 1) * TILE]];
    }
    lt.big_off = 0;
    tiles[t] = lt;
}

// exclusive scan of the big-group sizes (one workgroup); total[0] = sum
__global__ __launch_bounds__(1024) void loc_scan_kernel(LocTile* __restrict__ tiles, u32 ntiles, u32* __restrict__ total) {
    __shared__ u32 s_w[16];
    __shared__ u32 s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 base = 0; base < ntiles; base += 1024) {
        const u32 t = base + threadIdx.x;
        const u32 v = (t < ntiles) ? tiles[t].end - tiles[t].local_end : 0u;
        u32 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 x = __shfl_up(incl, o);
            if (lane >= o) incl += x;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        u32 off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        if (t < ntiles) tiles[t].big_off = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = s_carry;
}

// the big groups to / from their compact list (to_list: list <- records; else records <- list).  A big group is cut
// into gridDim.y slices, one workgroup each: word-like text has groups of 10^5..10^6 records (the most frequent words),
// and one workgroup copying such a group alone took longer than the sort of the whole list (1.5 of 13 ms, words 1e8).
__global__ __launch_bounds__(256) void loc_big_copy_kernel(const LocTile* __restrict__ tiles, u32 ntiles, bool to_list,
                                                           u64* __restrict__ keys, u32* __restrict__ vals,
                                                           u64* __restrict__ lkeys, u32* __restrict__ lvals) {
    for (u32 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const LocTile lt = tiles[t];
        const u32 cnt = lt.end - lt.local_end;
        if (cnt == 0) continue;
        const u32 per = (cnt + gridDim.y - 1) / gridDim.y;
        const u32 lo = blockIdx.y * per;
        const u32 hi = (lo + per < cnt) ? lo + per : cnt;
        for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const u64 r = (u64)lt.local_end + i, l = (u64)lt.big_off + i;
            if (to_list) { lkeys[l] = keys[r]; lvals[l] = vals[r]; }
            else { keys[r] = lkeys[l]; vals[r] = lvals[l]; }
        }
    }
}

struct LocSortArgs {
    const u64* keys_in;
    const u32* vals_in;
    u64* keys_out;
    u32* vals_out;
    const LocTile* tiles;
    int begin_bit;    // lowest significant key bit
    int gid_shift;    // key >> gid_shift = group id (64: there is no group id, one group)
    int top;          // one past the highest bit that can differ inside a tile: min(end_bit, gid_shift + LOC_GID_BITS)
    int passes;       // ceil((top - begin_bit) / 8)
};

// PACKED (the usual case: at most 52 key bits differ inside a tile): a record travels as ONE 64-bit word -- the
// significant key bits above the record's 12-bit position in the tile -- so a pass moves 8 instead of 12 bytes per
// record through LDS and the values are fetched once, at the end, from where they lie (36 KB of LDS: four workgroups
// per CU instead of three).
template <bool PACKED>
__global__ __launch_bounds__(LOC_BLOCK) void loc_sort_kernel(LocSortArgs a) {
    constexpr int WAVES = LOC_BLOCK / WAVE;
    __shared__ u64 s_key[LOC_CAP];
    __shared__ u32 s_val[PACKED ? 1 : LOC_CAP];
    __shared__ u32 s_whist[WAVES * RADIX];
    __shared__ u32 s_wsum[RADIX / WAVE];
    const LocTile lt = a.tiles[blockIdx.x];
    const u32 cnt = lt.local_end - lt.begin;   // <= LOC_CAP
    if (cnt == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 woff = (u32)wave * (WAVE * LOC_ITEMS) + lane;
    const u64* kin = a.keys_in + lt.begin;
    const u32* vin = a.vals_in + lt.begin;
    // the tile's smallest group id comes off every key: what is left of it fits LOC_GID_BITS
    const u64 base = (a.gid_shift < 64) ? ((kin[0] >> a.gid_shift) << a.gid_shift) : 0ull;
    const int first_bit = PACKED ? LOC_GID_BITS : a.begin_bit;               // lowest bit the passes look at
    const int top = PACKED ? a.top - a.begin_bit + LOC_GID_BITS : a.top;      // one past the highest

    u64 key[LOC_ITEMS];
    u32 val[PACKED ? 1 : LOC_ITEMS];
#pragma unroll
    for (int j = 0; j < LOC_ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        // padding sorts last in every pass and stays behind the records (stable)
        if (PACKED) key[j] = (p < cnt) ? ((((kin[p] - base) >> a.begin_bit) << LOC_GID_BITS) | (u64)p) : ~0ull;
        else {
            key[j] = (p < cnt) ? kin[p] - base : ~0ull;
            val[j] = (p < cnt) ? vin[p] : 0u;
        }
    }
    u32* wh = s_whist + wave * RADIX;
    for (int pass = 0; pass < a.passes; ++pass) {
        const int shift = first_bit + RADIX_BITS * pass;
        const int bits = (top - shift) < RADIX_BITS ? (top - shift) : RADIX_BITS;
        const u32 mask = (1u << bits) - 1u;
        for (int i = tid; i < WAVES * RADIX; i += LOC_BLOCK) s_whist[i] = 0;
        __syncthreads();
        u32 rd[LOC_ITEMS];
        wave_rank<true>(key, shift, mask, woff, LOC_CAP, wh, rd);
        __syncthreads();
        // digit counts of the tile -> per-wave exclusive offsets (the first RADIX threads: one digit each)
        u32 c = 0, incl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const u32 t = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = c;
                c += t;
            }
            incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) s_wsum[wave] = incl;
        }
        __syncthreads();
        if (tid < RADIX) {
            u32 excl = incl - c;
            for (int i = 0; i < wave; ++i) excl += s_wsum[i];
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += excl;
        }
        __syncthreads();
        // records to their sorted place in LDS (they live in registers: nothing in LDS is still needed), and back in
        // wave-striped order for the next pass
#pragma unroll
        for (int j = 0; j < LOC_ITEMS; ++j) {
            const u32 pos = wh[rd[j] >> 16] + (rd[j] & 0xFFFFu);
            s_key[pos] = key[j];
            if (!PACKED) s_val[pos] = val[j];
        }
        __syncthreads();
        if (pass + 1 < a.passes) {
#pragma unroll
            for (int j = 0; j < LOC_ITEMS; ++j) {
                key[j] = s_key[woff + j * WAVE];
                if (!PACKED) val[j] = s_val[woff + j * WAVE];
            }
            __syncthreads();
        }
    }
    u64* kout = a.keys_out + lt.begin;
    u32* vout = a.vals_out + lt.begin;
    for (u32 p = tid; p < cnt; p += LOC_BLOCK) {
        const u64 k = s_key[p];
        if (PACKED) {
            kout[p] = ((k >> LOC_GID_BITS) << a.begin_bit) + base;
            vout[p] = vin[(u32)k & (LOC_CAP - 1u)];   // the record's position in the tile before the sort
        } else {
            kout[p] = k + base;
            vout[p] = s_val[p];
        }
    }
}

static_assert(LOC_CAP == (1u << LOC_GID_BITS), "a position in the tile fits the bits the local group id frees");
static_assert(LOC_BLOCK >= RADIX && LOC_BLOCK % WAVE == 0, "one digit per thread in the scan");

}  // namespace sa
