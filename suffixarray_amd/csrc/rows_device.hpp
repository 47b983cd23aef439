// rows_device.hpp -- hits -> distinct rows on the device, for whole batches.
//
// Replaces, by function, the per-hit loop of get_matching_records_file (engine.c:1364-1388: for every hit seek to the
// enclosing row) and what the reference's Python layer does around it per query (suffix_array.pyx:221-247).  Until
// round 3 this library found all ranges of a batch in one launch and then looped PER QUERY over synchronous copies of
// SA slabs + a host binary search + a hash set (records.hpp: distinct_rows).  Here one workgroup per query walks the
// range SA[first ..] in chunks of 256 hits:
//     hit -> row            binary search over the row-start table in HBM
//     row -> first seen?    LDS hash table keyed by row, value = index of the row's FIRST hit (64-bit entries:
//                           compare-and-swap claims a slot, atomic min keeps the smallest hit index of a row)
//     new rows, in hit order: ballot prefix over the chunk -> out_rows[query][have ...]
// until k distinct rows are collected or the range ends -- exactly the rows, in exactly the order, that
// distinct_rows() returns (tests compare the two).  k <= ROWS_K_MAX; larger k ("all rows") stays on the host path.
//
// A workgroup per query keeps ~1000 queries in flight on the chip, each behind a chain of ~26 dependent probes: a batch of
// 1e7 patterns over the N = 1e9 text (27 % hit, nearly all exactly once) spent 42 ms there.  In a batch, ranges of at most
// ROWS_LANE_MAX hits -- misses included -- are therefore answered by ONE LANE each first (rows_lane_kernel: 64 independent
// probe chains per wave; the rows seen so far sit in registers); the longer ranges, appended to a list by wave-aggregated
// atomics, are walked by ONE WAVE each when k <= ROWS_WAVE_K_MAX (rows_wave_kernel) and otherwise -- or when a wave gives a
// range up -- by the workgroup form.  Same rows in the same order by construction (first-hit order).  1e6 sampled names of the
// config-5 column, k = 16: rows kernels 10.7 ms (workgroups only) -> 7.8 (lanes) -> 2.9 (lanes + waves).
#pragma once
#include "common.hpp"

namespace sa {

constexpr u32 ROWS_K_MAX = 4096;
constexpr int ROWS_COARSE_SHIFT = 8;
// (measured and not kept: a 1024-slot table for k <= 256 -- a table never holds more than k + 255 rows -- for eight workgroups
//  per CU instead of four or five: 1e6 names, k = 16: 14.7-14.9 ms against 13.9 with the 4096-slot table, profiles/r04_o_names_rows.log)
constexpr u32 ROWS_SLOTS_SMALL = 4096;    // k <= 1536: 32 KB of LDS
constexpr u32 ROWS_K_SMALL = 1536;
constexpr u32 ROWS_SLOTS_LARGE = 16384;   // k <= 4096: 128 KB of LDS (one workgroup per CU)
constexpr u32 ROWS_LANE_MAX = 4;          // hits of a range that one lane answers on its own (batches)
constexpr u64 ROWS_LANE_MIN_BATCH = 4096; // smaller batches go straight to the workgroup form (one launch)
constexpr u32 ROWS_WAVE_K_MAX = 64;       // batches, k <= 64: ranges of more than ROWS_LANE_MAX hits are walked by ONE WAVE each, 64 hits at a time
constexpr u32 ROWS_WAVE_MAX_HITS = 4096;  //  (a table per wave never holds more than k - 1 + 64 rows: 256 slots, 8 KB per workgroup of four waves --
constexpr u32 ROWS_WAVE_SLOTS = 256;      //   ~8000 ranges in flight on the chip instead of ~1100 workgroups, each behind the same chain of ~26 probes);
                                          //  a range whose first ROWS_WAVE_MAX_HITS hits do not yield k rows (many hits in few rows) is handed on to the
                                          //  workgroup form, which walks 256 hits at a time and starts it again

struct RowsArgs {
    const u32* sa;
    const sa_hip_pair_u32* ranges;   // [q], conventions of get_substring_positions (engine.c:896-898, 916-917)
    u64 q;
    const u64* row_starts;           // [num_rows], ascending, row_starts[0] = 0
    u64 num_rows;
    const u64* coarse;               // [coarse_n] = row_starts[j << ROWS_COARSE_SHIFT]: every 256th start (1.5 MB for 50M rows: stays in
    u64 coarse_n;                    //  the L2), or nullptr: a hit finds its row in 18 cached + 8 local probes instead of 26 HBM round trips
    u32 k;                           // 1 .. ROWS_K_MAX (and <= what SLOTS allows)
    u32* out_rows;                   // [q][k]
    u32* out_counts;                 // [q]
    const u32* pending = nullptr;    // rows_kernel / rows_wave_kernel: the queries left for it ([*n_pending], any order), or nullptr: all q
    const u32* n_pending = nullptr;
    u32* handoff = nullptr;          // rows_wave_kernel: the workgroup form's list and its length, for the ranges it gives up
    u32* n_handoff = nullptr;
};

// the row that holds text position pos: the last row whose start is <= pos (row_starts[0] = 0)
__device__ __forceinline__ u32 row_of_pos(const RowsArgs& a, const u64 pos) {
    u64 lo = 0, hi = a.num_rows;             // first row whose start is > pos
    if (a.coarse) {                          // (uniform) narrow [lo, hi) to one block of 256 rows through the small table
        u64 cl = 0, ch = a.coarse_n;         // first block whose first row starts > pos (>= 1: coarse[0] = 0 <= pos)
        while (cl < ch) {
            const u64 mid = (cl + ch) >> 1;
            if (a.coarse[mid] <= pos) cl = mid + 1; else ch = mid;
        }
        lo = (cl - 1) << ROWS_COARSE_SHIFT;  // row_starts[lo] <= pos
        const u64 end = cl << ROWS_COARSE_SHIFT;
        hi = end < a.num_rows ? end : a.num_rows;   // cl < coarse_n: row_starts[end] > pos
    }
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (a.row_starts[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    return (u32)(lo - 1);
}

__device__ __forceinline__ u32 hits_of_range(const sa_hip_pair_u32 rg) {   // miss: second = first - 1, or both UINT32_MAX
    return (rg.first != 0xFFFFFFFFu && (u32)(rg.second - rg.first + 1u) != 0u) ? rg.second - rg.first + 1u : 0u;
}

// One lane per query: ranges of <= ROWS_LANE_MAX hits are answered completely (distinct rows in first-hit order, at most k),
// the others are appended to pending[] (their out_counts entry is written by rows_kernel).
// pending[0 .. q) collects the ranges for the wave form (use_wave: k <= ROWS_WAVE_K_MAX), pending[q .. 2q) those for the workgroup form
// (without the wave form: all the others); n_pending[0] / [1] their lengths.
__global__ __launch_bounds__(256) void rows_lane_kernel(RowsArgs a, u32* __restrict__ pending, u32* __restrict__ n_pending, const int use_wave) {
    const u64 qi = (u64)blockIdx.x * 256 + threadIdx.x;
    bool big = false, mid = false;
    if (qi < a.q) {
        const sa_hip_pair_u32 rg = a.ranges[qi];
        const u32 count = hits_of_range(rg);
        if (count <= ROWS_LANE_MAX) {
            u32 r[ROWS_LANE_MAX];
            u32 have = 0;
            for (u32 i = 0; i < count && have < a.k; ++i) {
                const u32 row = row_of_pos(a, a.sa[(u64)rg.first + i]);
                bool seen = false;
#pragma unroll
                for (u32 j = 0; j < ROWS_LANE_MAX; ++j) seen |= (j < have && r[j] == row);
                if (!seen) {
#pragma unroll
                    for (u32 j = 0; j < ROWS_LANE_MAX; ++j) if (j == have) r[j] = row;   // (no dynamic register indexing)
                    a.out_rows[qi * a.k + have] = row;
                    ++have;
                }
            }
            a.out_counts[qi] = have;
        } else if (use_wave) {
            mid = true;
        } else {
            big = true;
        }
    }
    const int lane = threadIdx.x & 63;
    const u64 mm = __ballot(mid);
    if (mm) {
        const int leader = __ffsll((unsigned long long)mm) - 1;
        u32 base = 0;
        if (lane == leader) base = atomicAdd(&n_pending[0], (u32)__popcll(mm));
        base = __shfl(base, leader);
        if (mid) pending[base + (u32)__popcll(mm & lanemask_lt())] = (u32)qi;
    }
    const u64 m = __ballot(big);
    if (m) {
        const int leader = __ffsll((unsigned long long)m) - 1;
        u32 base = 0;
        if (lane == leader) base = atomicAdd(&n_pending[1], (u32)__popcll(m));
        base = __shfl(base, leader);
        if (big) pending[a.q + base + (u32)__popcll(m & lanemask_lt())] = (u32)qi;
    }
}

// LDS traffic of ONE wave ordered against itself (the waves of rows_wave_kernel run independent loops: no workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// One WAVE per range of the list: the workgroup form's walk (hit -> row, first-seen test in an LDS table, new rows in hit order by a
// ballot prefix) over 64 hits at a time with a table of the wave's own -- the same rows in the same order (a row's entry keeps the
// smallest hit index whatever the width of the chunks that brought its hits).
__global__ __launch_bounds__(256) void rows_wave_kernel(RowsArgs a) {
    __shared__ unsigned long long s_tab[4][ROWS_WAVE_SLOTS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long* tab = s_tab[wave];
    const u64 nq = (u64)*a.n_pending;
    for (u64 w = (u64)blockIdx.x * 4 + wave; w < nq; w += (u64)gridDim.x * 4) {
        const u64 qi = a.pending[w];
        const sa_hip_pair_u32 rg = a.ranges[qi];
        const u32 count = hits_of_range(rg);
        for (u32 s = lane; s < ROWS_WAVE_SLOTS; s += 64) tab[s] = 0ull;
        wave_lds_sync();
        u32 have = 0;
        const u32 walk = count < ROWS_WAVE_MAX_HITS ? count : ROWS_WAVE_MAX_HITS;
        for (u32 base = 0; base < walk && have < a.k; base += 64) {
            const u32 i = base + (u32)lane;
            const bool valid = i < count;
            u32 row = 0, slot = 0;
            if (valid) {
                row = row_of_pos(a, a.sa[(u64)rg.first + i]);
                const unsigned long long key = (unsigned long long)(row + 1u) << 32;
                slot = (row * 0x9E3779B1u) >> 24;
                while (true) {
                    const unsigned long long old = atomicCAS(&tab[slot], 0ull, key | i);
                    if (old == 0ull) break;
                    if ((old >> 32) == (key >> 32)) { atomicMin(&tab[slot], key | i); break; }
                    slot = (slot + 1u) & (ROWS_WAVE_SLOTS - 1u);
                }
            }
            wave_lds_sync();
            const bool win = valid && (u32)tab[slot] == i;   // this hit is the first one of its row
            const u64 m = __ballot(win);
            if (win) {
                const u32 p = have + (u32)__popcll(m & lanemask_lt());
                if (p < a.k) a.out_rows[qi * a.k + p] = row;
            }
            have += (u32)__popcll(m);
            wave_lds_sync();   // the reads above before the next chunk's atomics
        }
        if (have < a.k && walk < count) {       // (uniform) handed on: the workgroup form writes this range's rows and count
            if (lane == 0) a.handoff[atomicAdd(a.n_handoff, 1u)] = (u32)qi;
        } else if (lane == 0) {
            a.out_counts[qi] = have < a.k ? have : a.k;
        }
    }
}

// the rows of ONE range (all 256 threads of the workgroup; s_tab: SLOTS entries, s_wcnt: 4)
template <u32 SLOTS>
__device__ __forceinline__ void rows_of_range(const RowsArgs& a, const u64 qi, const sa_hip_pair_u32 rg, unsigned long long* s_tab, u32* s_wcnt) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int SHIFT = 32 - (SLOTS == 4096 ? 12 : 14);
    static_assert(SLOTS == 4096 || SLOTS == 16384, "table sizes");
    {
        const u32 count = hits_of_range(rg);
        for (u32 i = tid; i < SLOTS; i += 256) s_tab[i] = 0ull;
        __syncthreads();
        u32 have = 0;
        for (u64 base = 0; base < count && have < a.k; base += 256) {
            const bool valid = base + (u64)tid < count;
            const u32 i = (u32)(base + (u64)tid);
            u32 row = 0, slot = 0;
            if (valid) {
                row = row_of_pos(a, a.sa[(u64)rg.first + i]);
                const unsigned long long key = (unsigned long long)(row + 1u) << 32;
                slot = (row * 0x9E3779B1u) >> SHIFT;
                while (true) {
                    const unsigned long long old = atomicCAS(&s_tab[slot], 0ull, key | i);
                    if (old == 0ull) break;
                    if ((old >> 32) == (key >> 32)) { atomicMin(&s_tab[slot], key | i); break; }
                    slot = (slot + 1u) & (SLOTS - 1u);
                }
            }
            sync_lds();   // LDS atomics above
            const bool win = valid && (u32)s_tab[slot] == i;   // this hit is the first one of its row
            const u64 m = __ballot(win);
            if (lane == 0) s_wcnt[wave] = (u32)__popcll(m);
            __syncthreads();
            u32 before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const u32 c = s_wcnt[w]; if (w < wave) before += c; total += c; }
            if (win) {
                const u32 p = have + before + (u32)__popcll(m & lanemask_lt());
                if (p < a.k) a.out_rows[qi * a.k + p] = row;
            }
            have += total;
            __syncthreads();   // s_wcnt is rewritten by the next chunk
        }
        if (tid == 0) a.out_counts[qi] = have < a.k ? have : a.k;
        __syncthreads();       // the table is cleared for the next query
    }
}

template <u32 SLOTS>
__global__ __launch_bounds__(256) void rows_kernel(RowsArgs a) {
    __shared__ unsigned long long s_tab[SLOTS];   // (row + 1) << 32 | index of the row's first hit; 0 = empty
    __shared__ u32 s_wcnt[4];
    const u64 nq = a.pending ? (u64)*a.n_pending : a.q;
    for (u64 i = blockIdx.x; i < nq; i += gridDim.x) {
        const u64 qi = a.pending ? (u64)a.pending[i] : i;
        rows_of_range<SLOTS>(a, qi, a.ranges[qi], s_tab, s_wcnt);
    }
}

// ONE query, one launch (the latency path of get_matching_records_file): thread 0 searches, the workgroup then collects the rows
// of the range it found -- the search kernel and the rows kernel of the batch path without the launch in between.
// (sa_query.hpp is included before this header: query_one.)
template <bool NARROW, u32 SLOTS>
__global__ __launch_bounds__(256) void query_rows_one_kernel(QueryArgs qa, CodeMap map, RowsArgs ra) {
    __shared__ unsigned long long s_tab[SLOTS];
    __shared__ u32 s_wcnt[4];
    __shared__ u16 s_map[256];
    __shared__ sa_hip_pair_u32 s_rg;
    s_map[threadIdx.x] = map.code[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        const sa_hip_pair_u32 rg = query_one<NARROW, 2>(qa, s_map, 0);
        qa.out[0] = rg;
        s_rg = rg;
    }
    __syncthreads();
    rows_of_range<SLOTS>(ra, 0, s_rg, s_tab, s_wcnt);
}

// pend: device buffer of 2 * a.q + 2 u32 (the two lists of the queries the lane kernel leaves + their lengths), or nullptr / a small
// batch: every query goes through the workgroup form.  waves = false: no wave form (A/B, tests).
inline void launch_rows(hipStream_t stream, RowsArgs a, u32* pend = nullptr, bool waves = true, u32 wave_groups_per_cu = 8) {
    if (a.q == 0) return;
    if (pend && a.q >= ROWS_LANE_MIN_BATCH && a.q <= 0xFFFFFFFFull) {
        u32* n_pending = pend + 2 * a.q;
        const bool use_wave = waves && a.k <= ROWS_WAVE_K_MAX;
        (void)hipMemsetAsync(n_pending, 0, 8, stream);
        hipLaunchKernelGGL(rows_lane_kernel, dim3((u32)((a.q + 255) / 256)), dim3(256), 0, stream, a, pend, n_pending, use_wave ? 1 : 0);
        if (use_wave) {
            a.pending = pend;
            a.n_pending = n_pending;
            a.handoff = pend + a.q;
            a.n_handoff = n_pending + 1;
            const u64 gw = std::min<u64>((a.q + 3) / 4, 256u * (u64)wave_groups_per_cu);
            hipLaunchKernelGGL(rows_wave_kernel, dim3((u32)gw), dim3(256), 0, stream, a);
        }
        a.pending = pend + a.q;
        a.n_pending = n_pending + 1;
    }
    u64 g = a.q;
    if (a.k <= ROWS_K_SMALL) {
        if (g > 256u * 8u) g = 256u * 8u;
        hipLaunchKernelGGL(rows_kernel<ROWS_SLOTS_SMALL>, dim3((u32)g), dim3(256), 0, stream, a);
    } else {
        if (g > 256u * 2u) g = 256u * 2u;
        hipLaunchKernelGGL(rows_kernel<ROWS_SLOTS_LARGE>, dim3((u32)g), dim3(256), 0, stream, a);
    }
}

}  // namespace sa
