// radix_sort.hpp -- hand-written LSD radix sort of (u64 key, u32 value) records for gfx950.
//
// One-sweep structure: per 8-bit digit ONE kernel that reads every record once and writes it
// once; digit histograms come for free from the kernel that produced the input (the key
// generator for pass 0, the previous pass for every later pass).
//
// XCD-aware decomposition (MI355X: 8 XCDs, 256 CUs, ~1000 tiles in flight, a poll of another
// workgroup's status word is a ~1 us fabric round trip): a single decoupled look-back chain over
// all tiles costs more than moving the data (measured: 58 % of the pass, tools/sortbench.hip).
// The input of every pass is therefore cut into NCHUNK = 8 contiguous chunks of tiles with
//   * a digit histogram PER CHUNK  -> digit_base[chunk][digit] is known before the pass starts,
//   * a tile ticket PER CHUNK      -> tiles of a chunk start in order,
//   * a look-back chain PER CHUNK  -> 8 independent chains, each 1/8 as long and 1/8 as busy.
// A workgroup takes its tile from the chunk of the XCD it runs on (HW_REG_XCC_ID) and steals
// from the other chunks when its own is exhausted; placement changes speed only, any workgroup
// may process any tile.  Because pass p+1 reads the records in the order pass p wrote them, the
// per-chunk histogram of pass p+1 is accumulated BY pass p: while a tile's keys sit in LDS with
// their destination index known, the workgroup counts (destination chunk, next digit) pairs in
// LDS and flushes the non-zero bins with global atomics.
//
// Inside a pass each workgroup
//   1. takes the next tile of a chunk from that chunk's atomic ticket (tiles of a chunk start in
//      ticket order, so every predecessor is already resident -> look-back cannot deadlock),
//   2. loads its tile wave-striped (64 lanes x 8 B = 512 B per load instruction),
//   3. ranks its keys per wave with ballot match masks (wave_rank: 64-wide; per digit bit one
//      v_bfe_i32, one v_cmp and two v_bitop3) and per-wave digit counters in LDS -- no LDS
//      atomics, stable by construction,
//   4. publishes its 256 digit counts as 8-byte {epoch,flag,count} granules (one relaxed
//      agent-scope store each; the data IS the flag, cdna_hip_programming.md G16/R2), reorders
//      its keys through LDS, then resolves its exclusive prefix by decoupled look-back over the
//      predecessor tiles of its chunk, SA_LB_WINDOW polls in flight per lane (relaxed agent-scope
//      loads; every spin is bounded and sets DeviceStatus.error); only every 4th tile publishes
//      its inclusive prefix as well (SA_INCL_MASK),
//   5. streams keys (then values) out of LDS so that each digit's run leaves the CU as contiguous
//      global stores, counting the next pass's (chunk, digit) histogram on the way.
// Algorithmic bytes per pass over M records: 2*M*(8+4).
//
// This replaces, by function only, the bucket placement / induced-sorting scans of the
// reference's libsais (libsais.c:1542-1614, 2110-2141, 2942-2975): same output order, no
// shared code or structure.
#pragma once
#include "common.hpp"
#include <type_traits>

namespace sa {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int NCHUNK = 8;         // look-back chains per pass = XCDs
constexpr int SORT_ITEMS = 16;    // records per thread
constexpr int MAX_PASSES = 8;
constexpr u32 SPIN_LIMIT = 1u << 22;
#ifndef SA_INCL_MASK
#define SA_INCL_MASK 3u
#endif
#ifndef SA_LB_WINDOW
#define SA_LB_WINDOW 4
#endif

// tile status granule: [63:34] epoch | [33:32] flag | [31:0] count
constexpr u64 FLAG_AGG = 1, FLAG_INCL = 2;
__device__ __forceinline__ u64 pack_status(u32 epoch, u64 flag, u32 v) {
    return ((u64)epoch << 34) | (flag << 32) | (u64)v;
}

__device__ __forceinline__ u32 xcc_id() {
    u32 x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

// Geometry of one sort: chunk c = tiles [c*tpc, min((c+1)*tpc, tiles)).
struct SortGeom {
    u32 n;
    u32 tile;        // records per tile (power of two)
    u32 tile_shift;  // log2(tile)
    u32 tiles;       // total tiles
    u32 tpc;         // tiles per chunk
};
inline SortGeom make_geom(u32 n, u32 tile) {
    SortGeom g;
    g.n = n;
    g.tile = tile;
    g.tile_shift = 0;
    while ((1u << g.tile_shift) < tile) ++g.tile_shift;
    g.tiles = div_up(n, tile);
    g.tpc = div_up(g.tiles ? g.tiles : 1, NCHUNK);
    return g;
}
// chunk of a tile index: number of chunk boundaries at or below it (7 compares, exact, no division)
__host__ __device__ __forceinline__ u32 chunk_of_tile(u32 t, u32 tpc) {
    u32 c = 0;
#pragma unroll
    for (u32 k = 1; k < (u32)NCHUNK; ++k) c += (t >= k * tpc) ? 1u : 0u;
    return c;
}

struct SortPassArgs {
    const u64* keys_in;
    const u32* vals_in;   // nullptr: value = record position (iota), first pass of a build
    u64* keys_out;
    u32* vals_out;
    SortGeom g;
    int shift;
    u32 mask;
    int next_shift;         // digit of the next pass; < 0: this is the last pass
    u32 next_mask;
    const u32* digit_base;  // [NCHUNK][RADIX] exclusive global offsets of this pass
    u32* next_hist;         // [NCHUNK][RADIX] histogram of the next pass (zeroed by the host)
    u64* status;            // [tiles][RADIX]
    u32* ticket;            // [NCHUNK] tile tickets of this pass (zeroed by the host)
    u32 epoch;
    DeviceStatus* dstat;
    int home_mode;          // 0: home chunk = XCC id (product); 1: chunk 0; 2: blockIdx & 7 (tools/sortbench.hip)
    u32* keys_out32;        // NARROW kernels: the key leaves as (u32)(key >> narrow_shift) (radix_narrow.hpp)
    int narrow_shift;
    u32 incl_mask;          // a tile publishes its inclusive prefix only if (index in chunk & incl_mask) == incl_mask
};

// ---- pass-0 histogram: digit [shift, shift+8) per chunk of the INPUT order ---------------------------
// hist layout: [chunk][RADIX].  Every workgroup owns a contiguous range of tiles so that it crosses
// a chunk boundary at most a few times (LDS histogram flushed at each crossing).
__device__ __forceinline__ void hist_flush(u32* s_h, u32* hist, u32 chunk) {
    sync_lds();   // LDS atomics of the tile loop must have landed (see sync_lds)
    for (int d = threadIdx.x; d < RADIX; d += blockDim.x) {
        const u32 v = s_h[d];
        if (v) {
            atomicAdd(&hist[chunk * RADIX + d], v);
            s_h[d] = 0;
        }
    }
    sync_lds();
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const u64* __restrict__ keys, SortGeom g, int shift,
                                                         u32 mask, u32* __restrict__ hist) {
    __shared__ u32 s_h[RADIX];
    for (int i = threadIdx.x; i < RADIX; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    const u32 per = (g.tiles + gridDim.x - 1) / gridDim.x;
    const u32 t_lo = blockIdx.x * per;
    const u32 t_hi = (t_lo + per < g.tiles) ? t_lo + per : g.tiles;
    u32 cur_chunk = chunk_of_tile(t_lo, g.tpc);
    for (u32 t = t_lo; t < t_hi; ++t) {
        const u32 c = chunk_of_tile(t, g.tpc);
        if (c != cur_chunk) { hist_flush(s_h, hist, cur_chunk); cur_chunk = c; }
        const u64 base = (u64)t * g.tile;
        for (u32 l = threadIdx.x; l < g.tile; l += blockDim.x) {
            const u64 i = base + l;
            if (i < g.n) atomicAdd(&s_h[(u32)(keys[i] >> shift) & mask], 1u);
        }
    }
    if (t_lo < t_hi) hist_flush(s_h, hist, cur_chunk);
}

// digit_base[c][d] = sum_{d'<d} total[d'] + sum_{c'<c} cnt[c'][d]; one workgroup of 256
__global__ __launch_bounds__(256) void radix_scan_hist_kernel(const u32* __restrict__ hist, u32* __restrict__ base) {
    __shared__ u32 s_w[4];
    const int d = threadIdx.x, lane = d & 63, w = d >> 6;
    u32 cnt[NCHUNK];
    u32 c = 0;
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) { cnt[k] = hist[k * RADIX + d]; c += cnt[k]; }
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    u32 run = incl - c;
    for (int i = 0; i < w; ++i) run += s_w[i];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) { base[k * RADIX + d] = run; run += cnt[k]; }
}

// ---- decoupled look-back with a window of LB_WINDOW predecessors in flight -------------------------
// A granule only ever goes  not-ready -> AGG -> INCL  and both published forms stay valid for
// whoever read them, so window entries loaded early never go stale in a harmful way.
template <int LB_WINDOW = SA_LB_WINDOW>
__device__ __forceinline__ u32 lookback_prefix(const u64* __restrict__ status, u32 tile, u32 first_tile, u32 digit,
                                               u32 epoch, DeviceStatus* dstat) {
    u32 prefix = 0;
    int64_t t = (int64_t)tile - 1;
    const int64_t t0 = (int64_t)first_tile;
    while (t >= t0) {
        u64 w[LB_WINDOW];
#pragma unroll
        for (int i = 0; i < LB_WINDOW; ++i) {
            const int64_t ti = t - i;
            w[i] = (ti >= t0) ? __hip_atomic_load(&status[(u64)ti * RADIX + digit], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        bool done = false;
#pragma unroll
        for (int i = 0; i < LB_WINDOW; ++i) {
            const int64_t ti = t - i;
            if (!done && ti >= t0) {
                u64 x = w[i];
                u32 spins = 0;
                while (!((u32)(x >> 34) == epoch && ((x >> 32) & 3u) != 0)) {
                    ++spins;
                    if ((spins & 1023u) == 0) {
                        if (spins >= SPIN_LIMIT ||
                            __hip_atomic_load(&dstat->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            __hip_atomic_store(&dstat->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            return prefix;  // poisoned; the host reports SA_HIP_EINTERNAL
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                    x = __hip_atomic_load(&status[(u64)ti * RADIX + digit], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                prefix += (u32)x;
                if (((x >> 32) & 3u) == FLAG_INCL) done = true;
            }
        }
        if (done) break;
        t -= LB_WINDOW;
    }
    return prefix;
}

// ---- the pass -----------------------------------------------------------------------------------
// Written against the ISA: the first formulation of this kernel compiled to 2,800 vector-ALU
// instructions per wave and, with 16 waves per CU, was as much VALU-bound as HBM-bound.  This one
// needs about 1,400:
//   * ranking: per digit bit ONE v_bfe_i32 (e = -bit), ONE v_cmp (the ballot) and one three-input
//     bit operation (v_bitop3) per 32-bit half of the peer mask, peers &= ~(ballot ^ e); lane rank
//     with v_mbcnt; two records interleaved.  The per-wave digit counters are plain LDS words
//     (volatile generic pointers compiled to flat loads/stores with vmcnt(0) waits);
//   * the tile-local digit start is folded into the per-wave counters: one LDS read per record
//     when the keys are placed;
//   * full tiles run without any bounds checks (only the last tile of the input is partial);
//   * destination chunk of a record (for the next pass's histogram) from a per-digit
//     {base chunk, threshold} word instead of seven compares per record.
// ABL: ablation mask for tools/sortbench.hip only (0 in the product): 1 = no look-back (every tile
// of a chunk then writes to the same place: the stores stay in cache, NOT a bandwidth figure),
// 4 = no values, 8 = stores not scattered (streaming copy), 16 = no next-pass histogram.
// Per-wave stable ranking of ITEMS wave-striped records per lane by one digit: rd[j] = (number of records of
// this wave with the same digit that precede record j in memory order) | digit << 16; wh[digit] ends as
// the wave's digit count.  Two records at a time (two independent instruction streams); per digit bit
// ONE v_bfe_i32 (e = -bit), ONE v_cmp (the ballot) and one three-input bit operation per 32-bit half of
// the peer mask.  The empty asm statements keep the compiler from re-associating the mask updates into
// longer chains of two-input operations.
template <bool FULL, typename KeyT, int ITEMS>
__device__ __forceinline__ void wave_rank(const KeyT (&key)[ITEMS], int shift, u32 mask, u32 woff, u32 tile_n,
                                          u32* wh, u32 (&rd)[ITEMS]) {
    static_assert(ITEMS % 2 == 0, "records are ranked in pairs");
#pragma unroll
    for (int j = 0; j < ITEMS; j += 2) {
        bool valid[2];
        u32 d[2], lo[2], hi[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            valid[i] = FULL || (woff + (j + i) * WAVE) < tile_n;
            d[i] = (u32)(key[j + i] >> shift) & mask;
            lo[i] = ~0u; hi[i] = ~0u;
            if (!FULL) { const u64 vm = __ballot(valid[i]); lo[i] = (u32)vm; hi[i] = (u32)(vm >> 32); }
        }
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u32 e = (u32)__builtin_amdgcn_sbfe((int)d[i], b, 1);   // all ones where the bit is set
                asm("" : "+v"(e));
                const u64 m = __ballot(e != 0);
                lo[i] &= ~((u32)m ^ e);
                hi[i] &= ~((u32)(m >> 32) ^ e);
                asm("" : "+v"(lo[i]), "+v"(hi[i]));
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32 below = __builtin_amdgcn_mbcnt_hi(hi[i], __builtin_amdgcn_mbcnt_lo(lo[i], 0u));
            const u32 total = (u32)__popc(lo[i]) + (u32)__popc(hi[i]);
            const u32 prior = wh[d[i]];
            __builtin_amdgcn_wave_barrier();
            if (valid[i] && below == 0) wh[d[i]] = prior + total;
            __builtin_amdgcn_wave_barrier();
            rd[j + i] = (prior + below) | (d[i] << 16);
        }
    }
}

__device__ __forceinline__ u32 digit_of(u64 key, int shift, u32 mask) { return (u32)(key >> shift) & mask; }

template <bool FULL, int BLOCK, int ABL, bool NARROW>
__device__ __forceinline__ void onesweep_tile(const SortPassArgs& a, const u32 tile, const u32 chunk, const u32 tile_n,
                                              u64* s_keys, u32* s_whist, uint2* s_tab, u32* s_wsum) {
    constexpr int WAVES = BLOCK / WAVE;
    constexpr int TILE = BLOCK * SORT_ITEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 first_tile = chunk * a.g.tpc;
    const u64 tile_base = (u64)tile * TILE;
    const u32 woff = (u32)wave * (WAVE * SORT_ITEMS) + lane;   // tile-local index of this lane's record 0

    // this digit's global base for the chunk: requested now, needed after the look-back
    const u32 dbase = (tid < RADIX) ? a.digit_base[chunk * RADIX + tid] : 0u;

    // 1. load (wave-striped): wave w owns records [w*64*ITEMS, (w+1)*64*ITEMS) of the tile
    u64 key[SORT_ITEMS];
    const u64* kin = a.keys_in + tile_base;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        key[j] = (FULL || p < tile_n) ? kin[p] : ~0ull;
    }

    // 2. per-wave stable ranking with ballot match masks
    u32 rd[SORT_ITEMS];   // rank within (wave, digit) | digit << 16
    u32* wh = s_whist + wave * RADIX;
    wave_rank<FULL>(key, a.shift, a.mask, woff, tile_n, wh, rd);
    // values are fetched only now: their latency hides behind the count / look-back phase and
    // they do not occupy registers during ranking
    u32 val[SORT_ITEMS];
    if (a.vals_in) {
        const u32* vin = a.vals_in + tile_base;
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const u32 p = woff + j * WAVE;
            val[j] = (FULL || p < tile_n) ? vin[p] : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) val[j] = (u32)tile_base + woff + j * WAVE;
    }
    __syncthreads();

    // 3. tile digit counts -> publish aggregate -> exclusive scan over digits
    u32 count = 0, excl = 0;
    if (tid < RADIX) {
        u32 c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const u32 t = s_whist[w * RADIX + tid];
            s_whist[w * RADIX + tid] = c;   // exclusive over waves
            c += t;
        }
        count = c;
        __hip_atomic_store(&a.status[(u64)tile * RADIX + tid],
                           pack_status(a.epoch, tile == first_tile ? FLAG_INCL : FLAG_AGG, count),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += excl;   // tile-local start of (wave, digit)
    }
    __syncthreads();

    // 4. keys -> LDS at their tile-local sorted position (needs only tile-local offsets; gives the
    //    predecessor tiles time to publish before the look-back below)
    u32 pos[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        pos[j] = wh[rd[j] >> 16] + (rd[j] & 0xFFFFu);
        if (FULL || (woff + j * WAVE) < tile_n) s_keys[pos[j]] = key[j];
    }
    __syncthreads();   // keys are in LDS; s_whist is free from here on

    // 5. look-back: exclusive prefix of this tile's digits over the predecessor tiles of its chunk;
    //    meanwhile the other lanes clear the (chunk, next digit) histogram that reuses s_whist
    const bool has_next = (a.next_shift >= 0) && !(ABL & 16);
    if (has_next) for (int i = tid; i < NCHUNK * RADIX; i += BLOCK) s_whist[i] = 0;
    if (tid < RADIX) {
        u32 prefix = 0;
        if (tile > first_tile && !(ABL & 1)) {
            prefix = lookback_prefix(a.status, tile, first_tile, (u32)tid, a.epoch, a.dstat);
            // only every (incl_mask + 1)-th tile of a chunk publishes its inclusive prefix: a status store
            // is a fabric write of its own per lane (8-byte write-through), and the successors' look-back
            // is two to six tiles deep anyway (measured: every 4th tile 2.4-5.4 % faster than every tile)
            if (((tile - first_tile) & a.incl_mask) == a.incl_mask)
                __hip_atomic_store(&a.status[(u64)tile * RADIX + tid], pack_status(a.epoch, FLAG_INCL, prefix + count),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // this digit's run occupies global positions [g0, g0 + count); gdelta maps tile-local -> global
        const u32 g0 = dbase + prefix;
        const u32 gdelta = g0 - excl;
        // destination chunk of the run's records: c0, or c0 + 1 from tile-local position thr on
        // (a run is at most one tile long and a chunk at least one tile, so one boundary at most)
        const u32 c0 = chunk_of_tile(g0 >> a.g.tile_shift, a.g.tpc);
        const u64 bnd = (u64)(c0 + 1) * a.g.tpc << a.g.tile_shift;   // first record of chunk c0 + 1
        u32 thr = 0xFFFFu;
        if (c0 + 1 < (u32)NCHUNK && bnd < (u64)g0 + count) thr = (u32)(bnd - gdelta);
        s_tab[tid] = make_uint2(gdelta, (c0 << 16) | thr);
    }
    __syncthreads();

    // 6. coalesced global stores per digit run (+ next pass's per-chunk histogram); the uniform
    //    has_next decision is taken once so that the 16 LDS reads can be issued back to back
    u32 gidx[SORT_ITEMS];
    auto store_keys = [&](auto with_next) {
        constexpr bool NEXT = decltype(with_next)::value;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; ++k) {
            const u32 p = k * BLOCK + tid;
            if (FULL || p < tile_n) {
                const u64 kk = s_keys[p];
                const uint2 t = s_tab[digit_of(kk, a.shift, a.mask)];
                gidx[k] = t.x + p;
                if constexpr ((ABL & 8) != 0) gidx[k] = (u32)tile_base + p;
                if (NARROW) a.keys_out32[gidx[k]] = (u32)(kk >> a.narrow_shift);
                else a.keys_out[gidx[k]] = kk;
                if (NEXT) {
                    const u32 dn = digit_of(kk, a.next_shift, a.next_mask);
                    const u32 cn = (t.y >> 16) + (p >= (t.y & 0xFFFFu) ? 1u : 0u);
                    atomicAdd(&s_whist[cn * RADIX + dn], 1u);
                }
            }
        }
    };
    if (has_next) store_keys(std::true_type{}); else store_keys(std::false_type{});
    sync_lds();   // LDS atomics above (see sync_lds); every read of s_keys is done
    if (has_next) {
        for (int i = tid; i < NCHUNK * RADIX; i += BLOCK) {
            const u32 v = s_whist[i];
            if (v) atomicAdd(&a.next_hist[i], v);
        }
    }
    if constexpr ((ABL & 4) != 0) return;
    u32* s_vals = reinterpret_cast<u32*>(s_keys);
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j)
        if (FULL || (woff + j * WAVE) < tile_n) s_vals[pos[j]] = val[j];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) a.vals_out[gidx[k]] = s_vals[p];
    }
}

template <int BLOCK, int ABL = 0, bool NARROW = false>
__global__ __launch_bounds__(BLOCK, (BLOCK == 256) ? 3 : 4) void radix_onesweep_kernel(SortPassArgs a) {   // 256: LDS allows 3 workgroups of 4 waves
    constexpr int WAVES = BLOCK / WAVE;
    constexpr int TILE = BLOCK * SORT_ITEMS;
    constexpr int WH = (WAVES * RADIX > NCHUNK * RADIX) ? WAVES * RADIX : NCHUNK * RADIX;
    static_assert(BLOCK >= RADIX, "one thread per digit in the scan / look-back phase");
    static_assert(TILE < 0xFFFF, "tile-local positions must fit the 16-bit threshold field");
    __shared__ __attribute__((aligned(16))) u64 s_keys[TILE];  // reused as u32 values afterwards
    __shared__ u32 s_whist[WH];    // per-wave digit counters; later the (chunk, next digit) histogram
    __shared__ uint2 s_tab[RADIX]; // per digit {global - local offset, base chunk << 16 | threshold}
    __shared__ u32 s_wsum[RADIX / WAVE];
    __shared__ u32 s_tile;   // global tile index, or 0xFFFFFFFF = nothing left
    __shared__ u32 s_chunk;

    const int tid = threadIdx.x;
    if (tid == 0) {
        // One tile per workgroup.  Start at the chunk of this XCD; a chunk whose ticket has run past
        // its tile count is exhausted, move on to the next (ONE atomic per attempt, no pre-check:
        // every extra dependent round trip here delays the tile's first load by ~1-3 us).
        u32 tile = 0xFFFFFFFFu, chunk = 0;
        const u32 home = a.home_mode == 0 ? xcc_id() : (a.home_mode == 1 ? 0u : (blockIdx.x & 7u));
        for (int k = 0; k < NCHUNK; ++k) {
            const u32 c = (home + k) & (NCHUNK - 1);
            const u32 first = c * a.g.tpc;
            if (first >= a.g.tiles) continue;
            const u32 cnt = (a.g.tiles - first) < a.g.tpc ? (a.g.tiles - first) : a.g.tpc;
            const u32 t = atomicAdd(&a.ticket[c], 1u);
            if (t < cnt) { tile = first + t; chunk = c; break; }
        }
        s_tile = tile;
        s_chunk = chunk;
    }
    for (int i = tid; i < WAVES * RADIX; i += BLOCK) s_whist[i] = 0;
    __syncthreads();
    const u32 tile = s_tile;
    if (tile == 0xFFFFFFFFu) return;  // block-uniform
    const u32 chunk = s_chunk;
    const u64 rest = (u64)a.g.n - (u64)tile * TILE;
    if (rest >= (u64)TILE)
        onesweep_tile<true, BLOCK, ABL, NARROW>(a, tile, chunk, (u32)TILE, s_keys, s_whist, s_tab, s_wsum);
    else
        onesweep_tile<false, BLOCK, ABL, NARROW>(a, tile, chunk, (u32)rest, s_keys, s_whist, s_tab, s_wsum);
}

// ---- host driver ----------------------------------------------------------------------------------

// HIP-event stopwatch for the sort passes of a build, by kernel:
//   0 radix_onesweep_kernel<512>            (u64 key, u32 value in and out)
//   1 radix_onesweep_kernel<512, 0, true>   (top digit of a narrow sort: u64 key in, u32 key + u32 value out)
//   2 seg_onesweep_kernel<512, false>       (u32 key, u32 value in and out)
//   3 seg_onesweep_kernel<512, true>        (u32 key, u32 value in; u64 key, u32 value out)
constexpr int PASS_KINDS = 4;
struct EventTimer {
    static constexpr int CAP = 128;
    hipEvent_t ev[2 * CAP];
    int kind_of[CAP];
    int used = 0;
    bool ready = false;
    double total_ms = 0.0;
    u64 launches = 0;
    double kind_ms[PASS_KINDS] = {0, 0, 0, 0};
    u64 kind_launches[PASS_KINDS] = {0, 0, 0, 0};
    u64 kind_bytes[PASS_KINDS] = {0, 0, 0, 0};

    int init() {
        for (int i = 0; i < 2 * CAP; ++i) SA_HIP_CHECK(hipEventCreate(&ev[i]));
        ready = true;
        return 0;
    }
    void destroy() {
        if (!ready) return;
        for (int i = 0; i < 2 * CAP; ++i) (void)hipEventDestroy(ev[i]);
        ready = false;
    }
    void reset() {
        used = 0; total_ms = 0.0; launches = 0;
        for (int k = 0; k < PASS_KINDS; ++k) { kind_ms[k] = 0.0; kind_launches[k] = 0; kind_bytes[k] = 0; }
    }
    int flush() {
        for (int i = 0; i < used; ++i) {
            SA_HIP_CHECK(hipEventSynchronize(ev[2 * i + 1]));
            float ms = 0.f;
            SA_HIP_CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            total_ms += ms;
            kind_ms[kind_of[i]] += ms;
        }
        used = 0;
        return 0;
    }
    int start(hipStream_t s, int kind = 0) {
        if (used == CAP) { int rc = flush(); if (rc) return rc; }
        kind_of[used] = kind;
        SA_HIP_CHECK(hipEventRecord(ev[2 * used], s));
        return 0;
    }
    // bytes: algorithmic bytes of the launch (records x bytes read + written per record)
    int stop(hipStream_t s, u64 bytes = 0) {
        SA_HIP_CHECK(hipEventRecord(ev[2 * used + 1], s));
        kind_launches[kind_of[used]] += 1;
        kind_bytes[kind_of[used]] += bytes;
        ++used;
        ++launches;
        return 0;
    }
};

struct RadixWorkspace {
    u64* status = nullptr;     // [max_tiles][RADIX]
    u32* small = nullptr;      // tickets[MAX_PASSES][NCHUNK] | hist[MAX_PASSES][NCHUNK][RADIX] | base[NCHUNK][RADIX]
    DeviceStatus* dstat = nullptr;
    u32 max_tiles = 0;
    u32 epoch = 0;
    int block = 512;           // workgroup size of the pass kernel (tile = block * SORT_ITEMS)
    EventTimer timer;          // onesweep launches only
    u64 pass_records = 0;      // sum over passes of records moved
    u64 pass_bytes = 0;        // sum over passes of algorithmic bytes (records x bytes read + written per record)
    u64 passes = 0;
    // debugging aid: called after every pass with the pass's output (SA_HIP_DEBUG_ROUNDS)
    void (*debug_hook)(void* ctx, int pass, int npasses, int shift, u32 mask, const u64* kin, const u32* vin,
                       const u64* keys, const u32* vals, u32 n) = nullptr;
    void* debug_ctx = nullptr;

    int num_cus = 256;         // hipDeviceProp_t.multiProcessorCount
    u32 tile() const { return (u32)block * SORT_ITEMS; }
    // persistent workgroups: what one launch keeps resident (LDS-limited: 2 x 512 or 3 x 256 per CU)
    u32 persistent_grid() const { return (u32)num_cus * (block == 512 ? 2u : 3u); }
    u32* tickets() const { return small; }
    u32* hist(int pass) const { return small + MAX_PASSES * NCHUNK + (size_t)pass * NCHUNK * RADIX; }
    u32* base() const { return small + MAX_PASSES * NCHUNK + (size_t)MAX_PASSES * NCHUNK * RADIX; }
    // tickets + histograms (what a sort zeroes up front)
    static size_t zero_bytes() { return (size_t)(MAX_PASSES * NCHUNK + MAX_PASSES * NCHUNK * RADIX) * sizeof(u32); }
    static size_t small_bytes() { return zero_bytes() + (size_t)NCHUNK * RADIX * sizeof(u32); }

    int init(u64 n_max, int block_threads) {
        block = block_threads;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            num_cus = prop.multiProcessorCount;
        max_tiles = div_up(n_max ? n_max : 1, 256 * SORT_ITEMS);  // sized for the smaller tile
        SA_HIP_CHECK(hipMalloc(&status, (size_t)max_tiles * RADIX * sizeof(u64)));
        SA_HIP_CHECK(hipMemset(status, 0, (size_t)max_tiles * RADIX * sizeof(u64)));
        SA_HIP_CHECK(hipMalloc(&small, small_bytes()));
        SA_HIP_CHECK(hipMalloc(&dstat, sizeof(DeviceStatus)));
        SA_HIP_CHECK(hipMemset(dstat, 0, sizeof(DeviceStatus)));
        SA_HIP_CHECK(hipStreamSynchronize(nullptr));   // the fills above run on the null stream, which a non-blocking stream does not wait for
        epoch = 0;
        return timer.init();
    }
    void destroy() {
        if (status) (void)hipFree(status);
        if (small) (void)hipFree(small);
        if (dstat) (void)hipFree(dstat);
        status = nullptr; small = nullptr; dstat = nullptr;
        timer.destroy();
    }
    void reset_stats() { timer.reset(); pass_records = 0; pass_bytes = 0; passes = 0; }
};

struct SortPlan {
    int begin_bit, end_bit, npasses;
    u32 last_mask;
    SortGeom g;
    int shift(int p) const { return begin_bit + RADIX_BITS * p; }
    u32 mask(int p) const { return (p == npasses - 1) ? last_mask : (u32)(RADIX - 1); }
};
inline int make_plan(const RadixWorkspace& ws, u32 n, int begin_bit, int end_bit, SortPlan& p) {
    p.begin_bit = begin_bit;
    p.end_bit = end_bit;
    p.npasses = (end_bit - begin_bit + RADIX_BITS - 1) / RADIX_BITS;
    if (p.npasses > MAX_PASSES) return fail(SA_HIP_EINVAL, "radix sort: more than 8 passes");
    const int last_bits = end_bit - begin_bit - RADIX_BITS * (p.npasses - 1);
    p.last_mask = (1u << last_bits) - 1u;
    p.g = make_geom(n, ws.tile());
    if (p.g.tiles > ws.max_tiles) return fail(SA_HIP_EINVAL, "radix sort: workspace too small");
    return 0;
}

// Zero tickets + histograms; a producer kernel may then fill ws.hist(0) for the plan's pass 0.
inline int radix_prepare(RadixWorkspace& ws, hipStream_t stream) {
    SA_HIP_CHECK(hipMemsetAsync(ws.small, 0, RadixWorkspace::zero_bytes(), stream));
    return 0;
}

// Sort n records by key bits [begin_bit, end_bit), stable.  Buffers ping-pong A -> B -> A ...;
// on return *keys_res / *vals_res point at the buffers holding the result.  iota_vals: the
// first pass generates value = position instead of reading valsA.  hist_ready: the caller ran
// radix_prepare() and a producer kernel already filled ws.hist(0) for exactly this plan.
// pass0 (may be null): launches pass 0 in place of radix_onesweep_kernel -- a producer that makes the keys on the fly
// (radix_narrow.hpp: text_low_pass_kernel reads the text instead of a key array; keysA is then never touched).
struct Pass0Launcher {
    void (*launch)(void* ctx, hipStream_t stream, const SortPassArgs& a, u32 grid) = nullptr;
    void* ctx = nullptr;
    u32 bytes_per_record = 20;   // algorithmic bytes of that pass (read + written)
};
inline int radix_sort_pairs(RadixWorkspace& ws, hipStream_t stream, u64* keysA, u32* valsA, u64* keysB, u32* valsB,
                            u32 n, int begin_bit, int end_bit, bool iota_vals, bool hist_ready, u64** keys_res,
                            u32** vals_res, const Pass0Launcher* pass0 = nullptr) {
    *keys_res = keysA;
    *vals_res = valsA;
    if (n == 0 || end_bit <= begin_bit) {
        if (iota_vals && n) return fail(SA_HIP_EINVAL, "radix_sort_pairs: iota with zero passes");
        return 0;
    }
    SortPlan pl;
    int rc = make_plan(ws, n, begin_bit, end_bit, pl);
    if (rc) return rc;
    if (!hist_ready) {
        if ((rc = radix_prepare(ws, stream))) return rc;
        u32 hgrid = pl.g.tiles < 2048u ? pl.g.tiles : 2048u;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(hgrid), dim3(256), 0, stream, keysA, pl.g, pl.shift(0), pl.mask(0),
                           ws.hist(0));
    }

    u64* kin = keysA; u32* vin = valsA; u64* kout = keysB; u32* vout = valsB;
    for (int p = 0; p < pl.npasses; ++p) {
        if (++ws.epoch >= (1u << 30)) {  // epoch wrap: re-zero the granules once per 2^30 passes
            SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
            ws.epoch = 1;
        }
        hipLaunchKernelGGL(radix_scan_hist_kernel, dim3(1), dim3(256), 0, stream, ws.hist(p), ws.base());
        SortPassArgs a;
        a.keys_in = kin;
        a.vals_in = (p == 0 && iota_vals) ? nullptr : vin;
        a.keys_out = kout;
        a.vals_out = vout;
        a.g = pl.g;
        a.shift = pl.shift(p);
        a.mask = pl.mask(p);
        const bool last = (p == pl.npasses - 1);
        a.next_shift = last ? -1 : pl.shift(p + 1);
        a.next_mask = last ? 0u : pl.mask(p + 1);
        a.digit_base = ws.base();
        a.next_hist = last ? nullptr : ws.hist(p + 1);
        a.status = ws.status;
        a.ticket = ws.tickets() + p * NCHUNK;
        a.epoch = ws.epoch;
        a.dstat = ws.dstat;
        a.home_mode = 0;
        a.incl_mask = SA_INCL_MASK;
        a.keys_out32 = nullptr;
        a.narrow_shift = 0;
        if ((rc = ws.timer.start(stream, 0))) return rc;
        const u32 grid = pl.g.tiles;   // one tile per workgroup
        const bool by_producer = (p == 0 && pass0 && pass0->launch);
        if (by_producer)
            pass0->launch(pass0->ctx, stream, a, grid);
        else if (ws.block == 512)
            hipLaunchKernelGGL((radix_onesweep_kernel<512, 0>), dim3(grid), dim3(512), 0, stream, a);
        else
            hipLaunchKernelGGL((radix_onesweep_kernel<256, 0>), dim3(grid), dim3(256), 0, stream, a);
        const u64 pass_b = (u64)n * (by_producer ? pass0->bytes_per_record : (a.vals_in ? 24u : 20u));   // pass 0 of a build generates its values
        if ((rc = ws.timer.stop(stream, pass_b))) return rc;
        ws.pass_records += n;
        ws.pass_bytes += pass_b;
        ws.passes += 1;
        if (ws.debug_hook) ws.debug_hook(ws.debug_ctx, p, pl.npasses, a.shift, a.mask, kin, a.vals_in, kout, vout, n);
        u64* tk = kin; kin = kout; kout = tk;
        u32* tv = vin; vin = vout; vout = tv;
    }
    SA_HIP_CHECK(hipGetLastError());
    *keys_res = kin;
    *vals_res = vin;
    return 0;
}

}  // namespace sa
