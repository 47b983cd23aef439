// radix_split.hpp -- the narrow-record sort in THREE passes over the records instead of five (round 4).
//
// radix_narrow.hpp sorts keys of <= 40 bits as: top digit (8 bits, from the text) + four LSD passes over the 32-bit
// remainder inside the 256 buckets.  Every pass costs ~3.3 ps per record whatever its bytes, so what moves a
// build is fewer record-passes.  On near-random text -- the only text that takes this plan -- the buckets are even enough
// for an MSD continuation:
//
//   pass              reads              writes                                   bytes / record
//   top digit         text               u32 narrow key, u32 suffix               1 + 8         (text_top_pass_kernel, in its CLAIM form)
//   histogram         u32 key            hist[bucket][next 10 key bits]           4             (split_hist_kernel)
//   split             u32 key, u32 val   the same, grouped by the next rb <= 10   8 + 8         (seg_split_kernel)
//   local finish      u32 key, u32 val   sorted u32 key, u32 suffix (+ int64),    8 + 8 (+ 8)   (local_finish_kernel)
//                                        directory slice, tied slots
//
// After the split pass a SUB-BUCKET -- the records that share the top digit and the next rb key bits, 256 << rb of them --
// is contiguous and, when the text is as even as the plan assumes, holds at most LOCAL_CAP = 8192 records: one workgroup
// takes it into LDS whole and orders it completely by the remaining key bits (ties by suffix index, which is what a stable
// LSD sort from the identity leaves).  D1 at n = 1e9: 137 781 sub-buckets of <= 7 955 records.
//
//   * The split pass need not be stable (the local pass orders by (key, suffix) whatever order it finds), so it ranks
//     with one returning LDS atomic per record on ONE tile-wide counter array -- no per-wave histograms, which at 1024 bins
//     would not fit beside the tile -- and a tile CLAIMS its place inside a bucket's bin with one returning global atomic
//     per non-empty bin: no published counts, no look-back, nothing waits for another workgroup (ATOMIC; the look-back form,
//     two adjacent bins per thread, is kept for A/B: 4.25 against 3.05-3.45 ms at n = 1e9).  Geometry and tickets are
//     seg_onesweep_kernel's, with a plan of its own (tiles of 14 336 records).
//   * The local pass: bin = the next 12 key bits, 4096 counters of 16 bits packed two per LDS word; one returning LDS atomic
//     per record gives its place in the bin, a scan the bin starts; the records go to LDS as u64 (key << 32 | suffix) in bin
//     order, and a record's final place is its bin's start + the number of records of the bin that compare smaller (1.8
//     members on average at n = 1e9, two per step).  A bin with many members only costs time (members^2 comparisons), never
//     correctness.  A large form (1024 threads, sub-buckets of <= 16 384 records, one workgroup per CU) takes texts whose
//     sub-buckets outgrow 8192 at the finest level.
//   * FLAGS: the build's first flags pass is folded into the local pass -- the slice of the query directory that belongs to a
//     sub-bucket is its bin-start table, subsampled; tied slots are found in the store loop (the next slot's key from the next
//     lane's register) and staged per sub-bucket for lite_gather_kernel.
//   * Whether the plan applies, and with how many bits rb, is decided on the device's numbers: the histogram is taken over
//     hb = 10 key bits, split_levels_kernel gives the largest group for every rb <= hb, and the host -- the one
//     synchronisation this plan adds -- takes the smallest rb whose groups all fit LOCAL_CAP (D1: rb = 3 at n = 4.5e6,
//     10 at n = 1e9), else LOCAL_CAP_BIG; when none does the sort continues as radix_narrow.hpp's LSD passes (skewed text:
//     the histogram pass, and a second, stable run of the top-digit pass, are then the price of finding out).
//
// Replaces, by function only, the same libsais stages as radix_sort.hpp; no shared code or structure.
#pragma once
#include "flags_common.hpp"

namespace sa {

#ifndef SA_ABL
#define SA_ABL 0   // ablation mask of the FLAGS work in local_finish_kernel (tools/ only; 0 in the product)
#endif
constexpr int SPLIT_BITS = 10;
constexpr int SPLIT_NB = 1 << SPLIT_BITS;       // bins of the split pass (fewer are used when rb < SPLIT_BITS)
constexpr u32 LOCAL_CAP = 8192;                 // records of a sub-bucket the local pass holds: 512 threads x 16, 74 KB of LDS, two workgroups per CU
constexpr int LOCAL_BLOCK = 512;
constexpr int LOCAL_ITEMS = (int)(LOCAL_CAP / LOCAL_BLOCK);
constexpr u32 LOCAL_CAP_BIG = 16384;            // ... or, when no level fits that: 1024 threads x 16, 139 KB of LDS, one workgroup per CU
constexpr int LOCAL_BLOCK_BIG = 1024;
constexpr int LOCAL_BIN_BITS = 11;            // the key bits below a sub-bucket's that the plan asks for at least (bins of the local pass: 11 or 12 bits)
constexpr int SPLIT_HIST_COPIES = 4;

// histogram of the top rb bits of the narrow keys (digit = (key >> shift) & mask), per bucket: hist[b * SPLIT_NB + d]
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void split_hist_kernel(const u32* __restrict__ keys, const SegPlan* __restrict__ plan, int shift,
                                                           u32 mask, u32* __restrict__ hist, u32 tiles_per_block) {
    constexpr u32 TILE = BLOCK * ITEMS;
    constexpr int CS = SPLIT_NB + 1;
    __shared__ u32 s_h[SPLIT_HIST_COPIES * CS];
    __shared__ u32 s_t[RADIX + 1];
    for (int i = threadIdx.x; i <= RADIX; i += BLOCK) s_t[i] = plan->tprefix[i];
    for (int i = threadIdx.x; i < SPLIT_HIST_COPIES * CS; i += BLOCK) s_h[i] = 0;
    __syncthreads();
    const u32 F = s_t[RADIX];
    const u32 f_lo = blockIdx.x * tiles_per_block;
    const u32 f_hi = (f_lo + tiles_per_block < F) ? f_lo + tiles_per_block : F;
    if (f_lo >= f_hi) return;
    u32* my = s_h + (threadIdx.x & (SPLIT_HIST_COPIES - 1)) * CS;
    auto flush = [&](u32 bucket) {
        sync_lds();
        for (int d = threadIdx.x; d < SPLIT_NB; d += BLOCK) {
            u32 v = 0;
#pragma unroll
            for (int k = 0; k < SPLIT_HIST_COPIES; ++k) { v += s_h[k * CS + d]; s_h[k * CS + d] = 0; }
            if (v) atomicAdd(&hist[bucket * SPLIT_NB + d], v);
        }
        sync_lds();
    };
    u32 cur = seg_bucket_of(s_t, f_lo);
    for (u32 f = f_lo; f < f_hi; ++f) {
        const u32 b = seg_bucket_of(s_t, f);
        if (b != cur) { flush(cur); cur = b; }
        const u32 start = plan->bstart[b] + ((f - s_t[b]) * TILE);
        const u32 end = plan->bstart[b + 1];
        const u32 len = (end - start) < TILE ? (end - start) : TILE;
        if (len == TILE) {
            u32 k[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) k[j] = keys[start + j * BLOCK + threadIdx.x];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) atomicAdd(&my[(k[j] >> shift) & mask], 1u);
        } else {
            for (u32 l = threadIdx.x; l < len; l += BLOCK) atomicAdd(&my[(keys[start + l] >> shift) & mask], 1u);
        }
    }
    flush(cur);
}

// The histogram is taken over hb <= SPLIT_BITS key bits ("fine" bins); grouping the records by only the top k of those bits
// (level k) merges aligned runs of 2^(hb - k) fine bins.  levels[k] = the largest group at level k over all buckets, k = 0..hb:
// the host takes the SMALLEST level whose groups all fit the local pass (fewer, fuller sub-buckets and longer runs in the
// split pass).  One workgroup of SPLIT_NB threads per bucket.
__global__ __launch_bounds__(SPLIT_NB) void split_levels_kernel(const u32* __restrict__ hist, int hb, u32* __restrict__ levels) {
    __shared__ u32 s_p[SPLIT_NB + 1];
    __shared__ u32 s_w[SPLIT_NB / WAVE];
    const int b = blockIdx.x, d = threadIdx.x, lane = d & 63, w = d >> 6;
    const u32 c = hist[b * SPLIT_NB + d];
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    for (int i = 0; i < w; ++i) incl += s_w[i];
    s_p[d + 1] = incl;
    if (d == 0) s_p[0] = 0;
    __syncthreads();
    __shared__ u32 s_lv[SPLIT_BITS + 1][SPLIT_NB / WAVE];   // per level, the waves' maxima: ONE global atomic per level and bucket
    for (int k = 0; k <= hb; ++k) {                          // (one per level and wave: 45 000 atomics on 11 words, 80 us)
        const int g = hb - k;   // log2 of the fine bins per group
        u32 v = 0;
        if (d < (1 << k)) v = s_p[(d + 1) << g] - s_p[d << g];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const u32 t = __shfl_down(v, o); v = t > v ? t : v; }
        if (lane == 0) s_lv[k][w] = v;
    }
    __syncthreads();
    if (d <= hb) {
        u32 v = 0;
        for (int i = 0; i < SPLIT_NB / WAVE; ++i) v = s_lv[d][i] > v ? s_lv[d][i] : v;
        if (v) atomicMax(&levels[d], v);
    }
}

// For the level rb the host has chosen: base[b * SPLIT_NB + D] = bstart[b] + the records of bucket b whose top rb key bits are
// below D (the split pass's digit bases; D < 2^rb, the other entries 0), and the compact table of sub-bucket starts
// sub[(b << rb) + D] (sub[256 << rb] = n).
__global__ __launch_bounds__(SPLIT_NB) void split_scan_kernel(const u32* __restrict__ hist, const SegPlan* __restrict__ plan, int hb, int rb,
                                                              u32* __restrict__ base, u32* __restrict__ sub) {
    __shared__ u32 s_w[SPLIT_NB / WAVE];
    const int b = blockIdx.x, d = threadIdx.x, lane = d & 63, w = d >> 6;
    const u32 c = hist[b * SPLIT_NB + d];
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    u32 run = plan->bstart[b] + incl - c;
    for (int i = 0; i < w; ++i) run += s_w[i];
    const int g = hb - rb;
    u32 out = 0;
    if (d < (1 << hb) && (d & ((1 << g) - 1)) == 0) {
        out = run;
        sub[((u32)b << rb) + (u32)(d >> g)] = run;
    }
    // the split pass reads base[b][D] for D < SPLIT_NB: entry D = the start of group D
    __shared__ u32 s_o[SPLIT_NB];
    s_o[d] = 0;
    __syncthreads();
    if (d < (1 << hb) && (d & ((1 << g) - 1)) == 0) s_o[d >> g] = out;
    __syncthreads();
    base[b * SPLIT_NB + d] = s_o[d];
    if (d == 0 && b == RADIX - 1) sub[(u32)RADIX << rb] = plan->bstart[RADIX];
}

// decoupled look-back for the two adjacent bins a thread owns (status rows of SPLIT_NB granules); radix_sort.hpp's
// lookback_prefix with both chains advanced in the same window loads
template <int LB_WINDOW = SA_LB_WINDOW>
__device__ __forceinline__ void lookback_prefix_pair(const u64* __restrict__ status, u32 tile, u32 first_tile, u32 digit0,
                                                     u32 epoch, DeviceStatus* dstat, u32& p0, u32& p1) {
    u32 pre[2] = {0u, 0u};
    bool done[2] = {false, false};
    int64_t t = (int64_t)tile - 1;
    const int64_t t0 = (int64_t)first_tile;
    while (t >= t0 && !(done[0] && done[1])) {
        u64 w[LB_WINDOW][2];
#pragma unroll
        for (int i = 0; i < LB_WINDOW; ++i) {
            const int64_t ti = t - i;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                w[i][k] = (ti >= t0 && !done[k])
                              ? __hip_atomic_load(&status[(u64)ti * SPLIT_NB + digit0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int i = 0; i < LB_WINDOW; ++i) {
                const int64_t ti = t - i;
                if (!done[k] && ti >= t0) {
                    u64 x = w[i][k];
                    u32 spins = 0;
                    while (!((u32)(x >> 34) == epoch && ((x >> 32) & 3u) != 0)) {
                        ++spins;
                        if ((spins & 1023u) == 0) {
                            if (spins >= SPIN_LIMIT ||
                                __hip_atomic_load(&dstat->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                                __hip_atomic_store(&dstat->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                p0 = pre[0]; p1 = pre[1];
                                return;   // poisoned; the host reports SA_HIP_EINTERNAL
                            }
                        }
                        __builtin_amdgcn_s_sleep(1);
                        x = __hip_atomic_load(&status[(u64)ti * SPLIT_NB + digit0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    pre[k] += (u32)x;
                    if (((x >> 32) & 3u) == FLAG_INCL) done[k] = true;
                }
            }
        }
        t -= LB_WINDOW;
    }
    p0 = pre[0]; p1 = pre[1];
}

struct SplitPassArgs {
    const u32* keys_in;
    const u32* vals_in;
    u32* keys_out;
    u32* vals_out;
    const SegPlan* plan;
    int shift;             // digit = (key >> shift) & mask: the top rb bits of the narrow key
    u32 mask;
    const u32* digit_base; // [RADIX buckets][SPLIT_NB]
    u64* status;           // [flat tiles][SPLIT_NB]
    u32* ticket;           // [NCHUNK] (zeroed by the host)
    u32 epoch;
    DeviceStatus* dstat;
    u32 incl_mask;
    u32* cursor;           // ATOMIC form: records placed so far per (bucket, bin), zeroed by the host: cursor[bucket * cur_bs + bin * cur_ds]
    u32 cur_bs, cur_ds;    // ([bucket][bin]: SPLIT_NB, 1; SA_HIP_SPLIT_CURSOR_T=1: [bin][bucket] -- 1, RADIX -- so that the claims of a tile lie 1 KB
                           //  apart instead of in 3 KB of one bucket's row: measured no better, profiles/r04_claim_counters_layout.log)
};

// ATOMIC: a tile claims its place in a bin with one returning global atomic per non-empty bin instead of publishing its counts
// and looking back over its predecessors -- the pass need not be stable, so the tiles of a bucket need no order among
// themselves, and nothing waits for another workgroup.
template <bool FULL, int BLOCK, int ITEMS, bool ATOMIC>
__device__ __forceinline__ void split_tile(const SplitPassArgs& a, const u32 flat, const u32 first_flat, const u32 bucket,
                                           const u32 start, const u32 tile_n, u32* s_keys, u32* s_cnt, u32* s_gdelta, u32* s_wsum) {
    static_assert(SPLIT_NB == 2 * BLOCK, "two bins per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 woff = (u32)wave * (WAVE * ITEMS) + lane;

    // the bases of this thread's two bins: requested now, needed after the look-back
    const uint2 dbase = *reinterpret_cast<const uint2*>(a.digit_base + (size_t)bucket * SPLIT_NB + 2 * tid);

    // 1. load (wave-striped), 2. place inside the tile's bin from one returning LDS atomic per record (any order will do)
    u32 key[ITEMS];
    const u32* kin = a.keys_in + start;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        key[j] = (FULL || p < tile_n) ? kin[p] : 0u;
    }
    // (branch-free: a slot beyond the tile counts into the spare word s_cnt[SPLIT_NB] -- an atomic under a per-record branch made
    //  the compiler keep pos[] as one 32-wide vector value and spill it at every branch: 7 288 spilled registers)
    u32 pos[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 d = (key[j] >> a.shift) & a.mask;
        pos[j] = atomicAdd(&s_cnt[(FULL || (woff + j * WAVE) < tile_n) ? d : (u32)SPLIT_NB], 1u);
    }
    sync_lds();

    // 3. bin counts -> aggregate published -> exclusive scan over the bins
    const uint2 c = *reinterpret_cast<const uint2*>(s_cnt + 2 * tid);
    u32 claim0 = 0, claim1 = 0;
    if (ATOMIC) {   // requested now, needed for the stores
        u32* cur = a.cursor + (size_t)bucket * a.cur_bs + (size_t)(2 * tid) * a.cur_ds;
        if (c.x) claim0 = atomicAdd(cur, c.x);
        if (c.y) claim1 = atomicAdd(cur + a.cur_ds, c.y);
    } else {
        const u64 fl = (flat == first_flat) ? FLAG_INCL : FLAG_AGG;
        __hip_atomic_store(&a.status[(u64)flat * SPLIT_NB + 2 * tid], pack_status(a.epoch, fl, c.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.status[(u64)flat * SPLIT_NB + 2 * tid + 1], pack_status(a.epoch, fl, c.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const u32 tot = c.x + c.y;
    u32 incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();   // (every thread has read its two counters)
    u32 excl = incl - tot;
    for (int i = 0; i < wave; ++i) excl += s_wsum[i];
    *reinterpret_cast<uint2*>(s_cnt + 2 * tid) = make_uint2(excl, excl + c.x);   // the bins' tile-local starts
    __syncthreads();

    // 4. keys -> LDS at their tile-local position
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        pos[j] += s_cnt[(key[j] >> a.shift) & a.mask];
        if (FULL || (woff + j * WAVE) < tile_n) s_keys[pos[j]] = key[j];
    }
    __syncthreads();

    // 5. look-back inside the bucket, both bins of the thread at once
    {
        u32 p0 = claim0, p1 = claim1;
        if (!ATOMIC && flat > first_flat) {
            lookback_prefix_pair(a.status, flat, first_flat, 2u * (u32)tid, a.epoch, a.dstat, p0, p1);
            if (((flat - first_flat) & a.incl_mask) == a.incl_mask) {
                __hip_atomic_store(&a.status[(u64)flat * SPLIT_NB + 2 * tid], pack_status(a.epoch, FLAG_INCL, p0 + c.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.status[(u64)flat * SPLIT_NB + 2 * tid + 1], pack_status(a.epoch, FLAG_INCL, p1 + c.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        *reinterpret_cast<uint2*>(s_gdelta + 2 * tid) = make_uint2(dbase.x + p0 - excl, dbase.y + p1 - (excl + c.x));
    }
    __syncthreads();

    // 6. coalesced stores per bin run: keys, then the values through the same LDS
    //    (the tile-local positions wait packed two per register: with 32 records per thread the kernel would spill otherwise)
    static_assert(ITEMS % 2 == 0 && BLOCK * ITEMS <= 65536, "positions packed as 16-bit pairs");
    u32 pp[ITEMS / 2];
#pragma unroll
    for (int j = 0; j < ITEMS / 2; ++j) pp[j] = pos[2 * j] | (pos[2 * j + 1] << 16);
    u32 gidx[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) {
            const u32 kk = s_keys[p];
            gidx[k] = s_gdelta[(kk >> a.shift) & a.mask] + p;
            a.keys_out[gidx[k]] = kk;
        }
    }
    u32 val[ITEMS];
    const u32* vin = a.vals_in + start;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        val[j] = (FULL || p < tile_n) ? vin[p] : 0u;
    }
    __syncthreads();   // every read of s_keys is done
    u32* s_vals = s_keys;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (FULL || (woff + j * WAVE) < tile_n) s_vals[(pp[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu] = val[j];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) a.vals_out[gidx[k]] = s_vals[p];
    }
}

template <int BLOCK, int ITEMS, bool ATOMIC>
__global__ __launch_bounds__(BLOCK, 4) void seg_split_kernel(SplitPassArgs a) {
    constexpr u32 TILE = BLOCK * ITEMS;
    __shared__ __attribute__((aligned(16))) u32 s_keys[TILE];   // reused for the values
    __shared__ __attribute__((aligned(16))) u32 s_cnt[SPLIT_NB + 4];   // [SPLIT_NB]: spare word for slots beyond a partial tile
    __shared__ __attribute__((aligned(16))) u32 s_gdelta[SPLIT_NB];
    __shared__ u32 s_wsum[BLOCK / WAVE];
    __shared__ u32 s_t[RADIX + 1];
    __shared__ u32 s_b[RADIX + 1];
    __shared__ u32 s_c[NCHUNK + 1];
    __shared__ u32 s_flat;

    const int tid = threadIdx.x;
    u32 home = 0, t_home = 0;
    if (tid == 0) {
        home = xcc_id();
        t_home = atomicAdd(&a.ticket[home], 1u);
    }
    for (int i = tid; i <= RADIX; i += BLOCK) { s_t[i] = a.plan->tprefix[i]; s_b[i] = a.plan->bstart[i]; }
    if (tid <= NCHUNK) s_c[tid] = a.plan->cfirst[tid];
    for (int i = tid; i < SPLIT_NB + 4; i += BLOCK) s_cnt[i] = 0;
    __syncthreads();
    if (tid == 0) {
        u32 flat = 0xFFFFFFFFu;
        if (t_home < s_c[home + 1] - s_c[home]) flat = s_c[home] + t_home;
        for (int k = 1; k < NCHUNK && flat == 0xFFFFFFFFu; ++k) {   // own part exhausted: steal
            const u32 c = (home + k) & (NCHUNK - 1);
            const u32 cnt = s_c[c + 1] - s_c[c];
            if (cnt == 0) continue;
            const u32 t = atomicAdd(&a.ticket[c], 1u);
            if (t < cnt) flat = s_c[c] + t;
        }
        s_flat = flat;
    }
    __syncthreads();
    const u32 flat = s_flat;
    if (flat == 0xFFFFFFFFu) return;   // block-uniform
    const u32 bucket = seg_bucket_of(s_t, flat);
    const u32 first_flat = s_t[bucket];
    const u32 start = s_b[bucket] + (flat - first_flat) * TILE;
    const u32 rest = s_b[bucket + 1] - start;
    if (rest >= TILE) split_tile<true, BLOCK, ITEMS, ATOMIC>(a, flat, first_flat, bucket, start, TILE, s_keys, s_cnt, s_gdelta, s_wsum);
    else split_tile<false, BLOCK, ITEMS, ATOMIC>(a, flat, first_flat, bucket, start, rest, s_keys, s_cnt, s_gdelta, s_wsum);
}

// ---- local finish: one sub-bucket per workgroup, ordered completely in LDS ------------------------------------------------
struct LocalArgs {
    const u32* keys_in;
    const u32* vals_in;
    u32* keys_out;
    u32* vals_out;
    int64_t* vals_out64;   // may be null: the suffixes also leave as int64 (libsais64 layout, as seg_onesweep_kernel's LAST form)
    const u32* sub;        // [nsub + 1] starts of the sub-buckets
    int bin_shift;         // bin = (key >> bin_shift) & (2^BB - 1): the key bits right below the sub-bucket's
    DeviceStatus* dstat;
    // FLAGS: the build's first flags pass folded in (sa_build.hpp: flags_lite_kernel).  A sub-bucket holds every slot that
    // shares its 8 + rb key bits, so which slots are tied with a neighbour, and which directory buckets a slot owns, is
    // decided inside it: the pass over the sorted keys (4 n bytes read, 1.6 ms at n = 1e9) is not needed.
    DirArgs dir;           // the query path's bucket directory (dbits >= 8 + rb)
    uint2* counts;         // [sub-buckets] {active slots, active heads}, zeroed by the host
    LiteArgs lite;         // staging rows [sub-buckets][LITE_CAP] (lite.sa unused)
    int lo_shift;          // full key of a slot = (bucket << 56) | (narrow key << lo_shift)
    int rb;
    u32 n;
};

// What the caller of the sort provides when it wants the flags work done by the local pass: called once the level is known
// with the number of sub-buckets, fills dir / counts / lite (buffers sized for that many rows) -- or returns non-zero to decline.
struct LocalFlagsRequest {
    int (*prepare)(void* ctx, u32 nsub, LocalArgs* l) = nullptr;
    void* ctx = nullptr;
};

// BB: bin bits (11 or 12).  The counters are 16 bits wide, two per LDS word (a sub-bucket holds <= 8192 records), so that
// 4096 bins cost the 8 KB that 2048 32-bit counters did: 1.8 instead of 3.7 records per bin at n = 1e9, and it is the
// LARGEST bin among a wave's 64 records that sets the trip count of the counting loop.
template <int BB, bool FLAGS, int BLOCK = LOCAL_BLOCK>
__global__ __launch_bounds__(BLOCK, 4) void local_finish_kernel(LocalArgs a) {
    constexpr int ITEMS = LOCAL_ITEMS;
    constexpr u32 CAP = (u32)BLOCK * ITEMS;
    constexpr int NB = 1 << BB, WORDS = NB / 2, WPT = WORDS / BLOCK;   // packed counter words, words per thread in the scan
    static_assert(WPT >= 1 && WPT * BLOCK == WORDS, "the scan covers the counters exactly");
    static_assert(CAP < 65536, "16-bit counters and starts");
    __shared__ __attribute__((aligned(16))) u64 s_rec[CAP + 2];
    __shared__ __attribute__((aligned(16))) u32 s_cw[WORDS + 4];   // counts, then starts, of bins 2w | 2w + 1 << 16; [WORDS] low half: start[NB] = m
    __shared__ u32 s_wsum[BLOCK / WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 s = a.sub[blockIdx.x];
    const u32 m = a.sub[blockIdx.x + 1] - s;
    // FLAGS: the directory buckets of this sub-bucket's key prefix are [dfirst, dfirst + nd)
    const int g2 = FLAGS ? a.dir.dbits - 8 - a.rb : 0;
    const u32 dfirst = FLAGS ? (blockIdx.x << g2) : 0u;
    const u32 nd = FLAGS ? (1u << g2) : 0u;   // directory buckets of this sub-bucket (<= WORDS: the host has checked)
    if (FLAGS && tid == 0 && blockIdx.x == gridDim.x - 1) a.dir.dir[1u << a.dir.dbits] = a.n;   // the end marker
    if (m == 0) {
        if (FLAGS) {   // every bucket of an empty sub-bucket points at the next slot
            if (!(SA_ABL & 8)) for (u32 e = tid; e < nd; e += BLOCK) a.dir.dir[dfirst + e] = s;
        }
        return;
    }
    if (m > CAP) {   // cannot happen: the host has seen the largest sub-bucket
        if (tid == 0) __hip_atomic_store(&a.dstat->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const u32* kin = a.keys_in + s;
    const u32* vin = a.vals_in + s;
    if (!FLAGS && m == 1) {
        if (tid == 0) {
            const u32 k = kin[0], v = vin[0];
            a.keys_out[s] = k; a.vals_out[s] = v;
            if (a.vals_out64) a.vals_out64[s] = (int64_t)v;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) s_cw[WPT * tid + i] = 0;
    __syncthreads();

    // 1. load; place inside the bin from one returning LDS atomic per record
    u32 key[ITEMS], val[ITEMS], r[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = (u32)j * BLOCK + tid;
        key[j] = 0; val[j] = 0;
        if ((u32)j * BLOCK < m && p < m) { key[j] = kin[p]; val[j] = vin[p]; }
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = (u32)j * BLOCK + tid;
        r[j] = 0;
        if ((u32)j * BLOCK < m && p < m) {
            const u32 b = (key[j] >> a.bin_shift) & (u32)(NB - 1);
            const u32 sh = (b & 1u) * 16u;
            r[j] = (atomicAdd(&s_cw[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
        }
    }
    sync_lds();

    // 2. bin counts -> bin starts
    {
        u32 cw[WPT];
        u32 tot = 0;
#pragma unroll
        for (int i = 0; i < WPT; ++i) { cw[i] = s_cw[WPT * tid + i]; tot += (cw[i] & 0xFFFFu) + (cw[i] >> 16); }
        u32 incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        u32 run = incl - tot;
        for (int i = 0; i < wave; ++i) run += s_wsum[i];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const u32 c0 = cw[i] & 0xFFFFu, c1 = cw[i] >> 16;
            s_cw[WPT * tid + i] = run | ((run + c0) << 16);
            run += c0 + c1;
        }
        if (tid == BLOCK - 1) s_cw[WORDS] = m;
    }
    __syncthreads();
    const u16* s_st = reinterpret_cast<const u16*>(s_cw);   // start of bin b (little endian: the low half is the even bin)
    // FLAGS: dir[bkt] = first slot whose key's top bits are >= bkt = s + the records of the sub-bucket in lower buckets.  The bins
    // are the key bits right below the sub-bucket's and at least as fine as the directory (the host has checked g2 <= BB), so
    // that count is the start of the bucket's first bin: the slice of the directory is the bin-start table, subsampled (written at
    // the very end, behind the record stores: s_cw is not touched again)

    // 3. records -> LDS in bin order
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = (u32)j * BLOCK + tid;
        if ((u32)j * BLOCK < m && p < m) {
            const u32 slot = (u32)s_st[(key[j] >> a.bin_shift) & (u32)(NB - 1)] + r[j];
            s_rec[slot] = ((u64)key[j] << 32) | (u64)val[j];
        }
    }
    __syncthreads();

    // 4. final place of slot p: its bin's start + the records of the bin that compare smaller (key, then suffix);
    //    two records of the bin per step (one ds_read2_b64)
    u64 rec[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = (u32)j * BLOCK + tid;
        rec[j] = 0; r[j] = 0;
        if ((u32)j * BLOCK < m && p < m) {
            const u64 x = s_rec[p];
            const u32 bin = ((u32)(x >> 32) >> a.bin_shift) & (u32)(NB - 1);
            const u32 lo = s_st[bin], hi = s_st[bin + 1];
            u32 cnt = 0;
            for (u32 q = lo; q < hi; q += 2) {
                const u64 y0 = s_rec[q], y1 = s_rec[q + 1];   // (s_rec[m] may be read: padded, never counted)
                cnt += (y0 < x) ? 1u : 0u;
                cnt += (q + 1 < hi && y1 < x) ? 1u : 0u;
            }
            rec[j] = x;
            r[j] = lo + cnt;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = (u32)j * BLOCK + tid;
        if ((u32)j * BLOCK < m && p < m) s_rec[r[j]] = rec[j];
    }
    if (FLAGS && tid == 0) s_rec[m] = ~0ull;   // (the slot after the last one: never equal to a key)
    __syncthreads();

    // 5. out, coalesced: nothing but the loads and stores, unrolled -- this loop is what the pass's bandwidth hangs on (as a rolled
    //    loop over the slots: 4.9 -> 7.5 ms at n = 1e9; with a per-slot flags body inside its sixteen copies: 6.8-7.3 ms; a second
    //    walk over the sorted keys after it: 6.2 ms).  FLAGS asks one more thing of it: is the NEXT slot's key the same?  (the
    //    neighbour comes with the same LDS instruction.)  Only then -- 0.2 % of the slots of a near-random text -- the slot looks
    //    back as well and stages what is tied: itself when it starts the group, its successor always -- after the loop, from a
    //    bit mask (the staging code inside the sixteen copies cost 0.8 ms).  The rows of a sub-bucket arrive in any order;
    //    lite_gather_kernel puts a row into slot order.
    // FLAGS work that does not depend on the store loop goes BEFORE it -- whatever follows the stores keeps a finished workgroup
    // resident (measured: the same work behind the loop, in any arrangement, cost 0.8 ms at n = 1e9)
    auto stage_pair = [&](u32 p) {   // slots p and p + 1 carry the same key
        const u64 x = s_rec[p], xn = s_rec[p + 1];
        const bool starts = (p == 0) || ((u32)(s_rec[p - 1] >> 32) != (u32)(x >> 32));
        // (the sub-bucket's {active, heads} pair lives in global memory, zeroed by the host: the few threads that get here add to
        //  it, and the kernel needs neither a counter in LDS nor a barrier at its end)
        u32 at = atomicAdd(&a.counts[blockIdx.x].x, starts ? 2u : 1u);
        if (at + (starts ? 2u : 1u) > LITE_CAP) __hip_atomic_store(a.lite.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (starts) {
            atomicAdd(&a.counts[blockIdx.x].y, 1u);
            if (at < LITE_CAP) {
                const u64 row = (u64)blockIdx.x * LITE_CAP + at;
                a.lite.st_pos[row] = s + p; a.lite.st_idx[row] = (u32)x; a.lite.st_head[row] = 1;
            }
            ++at;
        }
        if (at < LITE_CAP) {
            const u64 row = (u64)blockIdx.x * LITE_CAP + at;
            a.lite.st_pos[row] = s + p + 1; a.lite.st_idx[row] = (u32)xn; a.lite.st_head[row] = 0;
        }
    };
    if (FLAGS) {
        if (!(SA_ABL & 1)) for (u32 e = tid; e < nd; e += BLOCK) a.dir.dir[dfirst + e] = s + (u32)s_st[e << (BB - g2)];
        // the pairs across a row of 16 lanes (slots 16 t + 15 | 16 t + 16)
        if (!(SA_ABL & 4))
        for (u32 p = (u32)tid * 16u + 15u; p + 1 < m; p += (u32)BLOCK * 16u)
            if ((u32)(s_rec[p] >> 32) == (u32)(s_rec[p + 1] >> 32)) stage_pair(p);
    }
    u32 tmask = 0;   // FLAGS: bit j = the slot after item j's carries the same key
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = (u32)j * BLOCK + tid;
        if ((u32)j * BLOCK < m) {   // (workgroup-uniform; a lane beyond m reads a stale word of s_rec and stores nothing)
            const u64 x = s_rec[p];
            const u32 k = (u32)(x >> 32);
            if (p < m) {
                a.keys_out[s + p] = k;
                a.vals_out[s + p] = (u32)x;
                if (a.vals_out64) a.vals_out64[s + p] = (int64_t)(u32)x;
            }
            if (FLAGS && !(SA_ABL & 2)) {
                // the next slot's key from the next lane's register (row_shl:1 -- no LDS read in this loop); the last lane of a
                // row of 16 sees its own key inverted, its pair has been looked at before the loop
                const u32 kn = (u32)__builtin_amdgcn_update_dpp((int)~k, (int)k, 0x101, 0xF, 0xF, false);
                tmask |= (kn == k && p + 1 < m) ? (1u << j) : 0u;
            }
        }
    }
    if (FLAGS) {
        while (tmask) {   // 3 % of the threads of a near-random text get here, with one bit
            const u32 j = (u32)__builtin_ctz(tmask);
            tmask &= tmask - 1u;
            stage_pair(j * BLOCK + (u32)tid);
        }
    }
}

}  // namespace sa
