// radix_narrow.hpp -- sort of (u64 key, iota value) records whose keys have at most 40 significant bits
// (bits [begin_bit, 64), begin_bit >= 24), in 8-byte instead of 12-byte records.
//
// An LSD sort has to carry the whole key through every pass.  If the TOP digit is sorted first
// instead (one stable pass of the ordinary one-sweep kernel, radix_sort.hpp), a record's top digit is
// given by where it lies -- its bucket -- and need not be stored any more: the other <= 32 key bits fit a
// u32, the record shrinks from 8 + 4 to 4 + 4 bytes, and the remaining digits are sorted by LSD passes
// that keep every record inside its bucket (256 independent sorts run side by side):
//
//   pass            reads            writes           bytes / record
//   top digit       text (1 byte)    u32 low key, u32 value (iota)     1 + 8    (text_top_pass_kernel; from a u64 key
//                                                                                array instead: 8 + 8, radix_onesweep_kernel<.., true>)
//   histogram       u32 low key      -                                 4        (digit 0 per bucket)
//   narrow passes   u32 key, u32 val u32 key, u32 val                  8 + 8
//   last pass       u32 key, u32 val u64 key (rebuilt), u32 val        8 + 12
//
// 40-bit keys (the build's choice for near-random text: 8 characters of 5 bits): 1 (top-digit histogram) + 9 + 4 +
// 3*16 + 20 = 82 bytes per character instead of 9 (key generation) + 20 + 4*24 = 125.
//
// Geometry of the narrow passes: bucket b = records [bstart[b], bstart[b+1]) of the arrays; its tiles
// start at the bucket start (no tile straddles two buckets, the last tile of a bucket is partial).
// Tiles are numbered bucket-major ("flat" index); tprefix[b] = flat index of bucket b's first tile.
// The flat range is cut into NCHUNK parts with one ticket each (a workgroup starts at the part of the XCD it
// runs on, as in radix_sort.hpp; the parts consist of whole buckets, so a chain never leaves its part and
// every predecessor of a running tile has started); the look-back chain of a tile is its BUCKET (tiles tprefix[b] .. flat-1),
// so there are up to 256 short chains instead of 8 long ones, and the digit bases are per bucket:
// base[b][d] = bstart[b] + sum_{d' < d} hist[b][d'].  Since records never leave their bucket, the histogram of
// the next pass is one LDS histogram per tile added to hist_next[b][.].
#pragma once
#include "radix_sort.hpp"
#include <vector>

namespace sa {

constexpr int NARROW_MAX_PASSES = 6;   // 4 passes of a 32-bit remainder; 2 + 4 of a 48-bit one (radix_narrow48.hpp)
constexpr int SEG_ITEMS = 24;   // records per thread of the narrow passes: tiles of 512 x 24 = 12288 records

struct SegPlan {              // device resident, written by seg_plan_kernel
    u32 bstart[RADIX + 1];    // first record of bucket b; [RADIX] = n
    u32 tprefix[RADIX + 1];   // flat index of the first tile of bucket b; [RADIX] = number of tiles
    u32 cfirst[NCHUNK + 1];   // ticket ranges: part c = flat tiles [cfirst[c], cfirst[c+1]), whole buckets each
};

// bucket sizes from the per-chunk histogram of the top digit -> SegPlan (one workgroup of 256)
__global__ __launch_bounds__(256) void seg_plan_kernel(const u32* __restrict__ hist_top, u32 tile, SegPlan* __restrict__ plan) {
    __shared__ u32 s_w[2][4];
    __shared__ u32 s_tp[RADIX + 1];
    const int d = threadIdx.x, lane = d & 63, w = d >> 6;
    u32 size = 0;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) size += hist_top[c * RADIX + d];
    const u32 tiles = (u32)(((u64)size + tile - 1) / tile);
    u32 is = size, it = tiles;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 a = __shfl_up(is, o), b = __shfl_up(it, o);
        if (lane >= o) { is += a; it += b; }
    }
    if (lane == 63) { s_w[0][w] = is; s_w[1][w] = it; }
    __syncthreads();
    u32 os = 0, ot = 0;
    for (int i = 0; i < w; ++i) { os += s_w[0][i]; ot += s_w[1][i]; }
    plan->bstart[d] = os + is - size;
    plan->tprefix[d] = ot + it - tiles;
    s_tp[d] = ot + it - tiles;
    if (d == RADIX - 1) { plan->bstart[RADIX] = os + is; plan->tprefix[RADIX] = ot + it; s_tp[RADIX] = ot + it; }
    __syncthreads();
    if (d <= NCHUNK) {
        // part c starts at the first bucket whose first tile lies at or beyond c/NCHUNK of all tiles
        const u32 F = s_tp[RADIX];
        const u32 target = (u32)(((u64)F * (u32)d) / NCHUNK);
        u32 lo = 0, hi = RADIX;   // first b in [0, RADIX] with tprefix[b] >= target
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (s_tp[mid] < target) lo = mid + 1; else hi = mid;
        }
        plan->cfirst[d] = (d == NCHUNK) ? F : s_tp[lo];
    }
}

// bucket of a flat tile index: the last b with tprefix[b] <= f (empty buckets share their successor's prefix)
__device__ __forceinline__ u32 seg_bucket_of(const u32* tprefix, u32 f) {
    u32 lo = 0, hi = RADIX;   // answer in [lo, hi)
#pragma unroll
    for (int s = 0; s < RADIX_BITS; ++s) {
        const u32 mid = (lo + hi) >> 1;
        if (tprefix[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

// histogram of digit [shift, shift + 8) of the narrow keys, per bucket: hist[b][d]; the LDS histogram is kept in
// SEG_HIST_COPIES lane-selected copies (fewer lanes of a wave on the same counter)
constexpr int SEG_HIST_COPIES = 4;
template <int BLOCK, int ITEMS, typename KT = u32>
__global__ __launch_bounds__(BLOCK) void seg_hist_kernel(const KT* __restrict__ keys, const SegPlan* __restrict__ plan, int shift,
                                                         u32 mask, u32* __restrict__ hist, u32 tiles_per_block) {
    constexpr u32 TILE = BLOCK * ITEMS;
    constexpr int CS = RADIX + 1;
    __shared__ u32 s_h[SEG_HIST_COPIES * CS];
    __shared__ u32 s_t[RADIX + 1];
    for (int i = threadIdx.x; i <= RADIX; i += BLOCK) s_t[i] = plan->tprefix[i];
    for (int i = threadIdx.x; i < SEG_HIST_COPIES * CS; i += BLOCK) s_h[i] = 0;
    __syncthreads();
    const u32 F = s_t[RADIX];
    const u32 f_lo = blockIdx.x * tiles_per_block;
    const u32 f_hi = (f_lo + tiles_per_block < F) ? f_lo + tiles_per_block : F;
    if (f_lo >= f_hi) return;
    u32* my = s_h + (threadIdx.x & (SEG_HIST_COPIES - 1)) * CS;
    auto flush = [&](u32 bucket) {
        sync_lds();
        for (int d = threadIdx.x; d < RADIX; d += BLOCK) {
            u32 v = 0;
#pragma unroll
            for (int k = 0; k < SEG_HIST_COPIES; ++k) { v += s_h[k * CS + d]; s_h[k * CS + d] = 0; }
            if (v) atomicAdd(&hist[bucket * RADIX + d], v);
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
        if (len == TILE) {   // all loads of the tile in flight before the first LDS atomic
            u32 k[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) k[j] = (u32)keys[start + j * BLOCK + threadIdx.x];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) atomicAdd(&my[(k[j] >> shift) & mask], 1u);
        } else {
            for (u32 l = threadIdx.x; l < len; l += BLOCK) atomicAdd(&my[((u32)keys[start + l] >> shift) & mask], 1u);
        }
    }
    flush(cur);
}

// base[b][d] = bstart[b] + sum_{d' < d} hist[b][d']; one workgroup of 256 per bucket
__global__ __launch_bounds__(256) void seg_scan_kernel(const u32* __restrict__ hist, const SegPlan* __restrict__ plan,
                                                       u32* __restrict__ base) {
    __shared__ u32 s_w[4];
    const int b = blockIdx.x, d = threadIdx.x, lane = d & 63, w = d >> 6;
    const u32 c = hist[b * RADIX + d];
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
    base[b * RADIX + d] = run;
}

struct SegPassArgs {
    const u32* keys_in;
    const u32* vals_in;
    u32* keys_out;         // narrow keys (passes before the last)
    u64* keys_out64;       // LAST: (bucket << 56) | (narrow key << lo_shift); nullptr: the last pass too leaves narrow keys
                           // in keys_out (the caller keeps them as they are: Builder::qkeys32)
    u32* vals_out;
    int64_t* vals_out64;   // LAST only, may be null: the values also leave as int64 (libsais64 layout: the widening pass of
                           // libsais64.c:6248-6259 folded into the sort's last store, 8 more bytes written per record
                           // instead of a separate 4 + 8 byte pass)
    const SegPlan* plan;
    int shift;             // digit of the narrow key
    u32 mask;
    int next_shift;        // < 0: last pass
    u32 next_mask;
    const u32* digit_base; // [RADIX buckets][RADIX]
    u32* next_hist;        // [RADIX buckets][RADIX] (zeroed by the host)
    u64* status;           // [flat tiles][RADIX]
    u32* ticket;           // [NCHUNK] (zeroed by the host)
    u32 epoch;
    DeviceStatus* dstat;
    int lo_shift;
    u32 incl_mask;
};

template <bool FULL, int BLOCK, int ITEMS, bool LAST, bool LATEV>
__device__ __forceinline__ void seg_tile(const SegPassArgs& a, const u32 flat, const u32 first_flat, const u32 bucket,
                                         const u32 start, const u32 tile_n, u32* s_keys, u32* s_whist, u32* s_gdelta,
                                         u32* s_wsum) {
    constexpr int WAVES = BLOCK / WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 woff = (u32)wave * (WAVE * ITEMS) + lane;

    // this digit's base inside the bucket: requested now, needed after the look-back
    const u32 dbase = (tid < RADIX) ? a.digit_base[bucket * RADIX + tid] : 0u;

    // 1. load (wave-striped)
    u32 key[ITEMS];
    const u32* kin = a.keys_in + start;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        key[j] = (FULL || p < tile_n) ? kin[p] : ~0u;
    }
    // 2. rank
    u32 rd[ITEMS];
    u32* wh = s_whist + wave * RADIX;
    wave_rank<FULL>(key, a.shift, a.mask, woff, tile_n, wh, rd);
    // values: right after the ranking (their latency hides behind the count / look-back phase), or -- LATEV -- only
    // when they are needed, which keeps 16 registers free through the look-back (three workgroups per CU instead of two)
    u32 val[ITEMS];
    const u32* vin = a.vals_in + start;
    auto load_vals = [&]() {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 p = woff + j * WAVE;
            val[j] = (FULL || p < tile_n) ? vin[p] : 0u;
        }
    };
    if (!LATEV) load_vals();
    __syncthreads();

    // 3. tile digit counts -> aggregate -> exclusive scan over digits
    u32 count = 0, excl = 0;
    if (tid < RADIX) {
        u32 c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const u32 t = s_whist[w * RADIX + tid];
            s_whist[w * RADIX + tid] = c;
            c += t;
        }
        count = c;
        __hip_atomic_store(&a.status[(u64)flat * RADIX + tid],
                           pack_status(a.epoch, flat == first_flat ? FLAG_INCL : FLAG_AGG, count),
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
        for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += excl;
    }
    __syncthreads();

    // 4. keys -> LDS at their tile-local sorted position
    u32 pos[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        pos[j] = wh[rd[j] >> 16] + (rd[j] & 0xFFFFu);
        if (FULL || (woff + j * WAVE) < tile_n) s_keys[pos[j]] = key[j];
    }
    __syncthreads();

    // 5. look-back inside the bucket; the other lanes clear the next-digit histogram (reuses s_whist)
    const bool has_next = !LAST;
    if (has_next) for (int i = tid; i < RADIX; i += BLOCK) s_whist[i] = 0;
    if (tid < RADIX) {
        u32 prefix = 0;
        if (flat > first_flat) {
            prefix = lookback_prefix(a.status, flat, first_flat, (u32)tid, a.epoch, a.dstat);
            if (((flat - first_flat) & a.incl_mask) == a.incl_mask)
                __hip_atomic_store(&a.status[(u64)flat * RADIX + tid], pack_status(a.epoch, FLAG_INCL, prefix + count),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gdelta[tid] = dbase + prefix - excl;
    }
    __syncthreads();

    // 6. coalesced stores per digit run
    u32 gidx[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) {
            const u32 kk = s_keys[p];
            gidx[k] = s_gdelta[(kk >> a.shift) & a.mask] + p;
            if (LAST) {
                if (a.keys_out64) a.keys_out64[gidx[k]] = ((u64)bucket << 56) | ((u64)kk << a.lo_shift);   // uniform branch
                else a.keys_out[gidx[k]] = kk;
            } else {
                a.keys_out[gidx[k]] = kk;
                atomicAdd(&s_whist[(kk >> a.next_shift) & a.next_mask], 1u);
            }
        }
    }
    sync_lds();   // LDS atomics above; every read of s_keys is done
    if (has_next) {
        for (int i = tid; i < RADIX; i += BLOCK) {
            const u32 v = s_whist[i];
            if (v) atomicAdd(&a.next_hist[bucket * RADIX + i], v);
        }
    }
    if (LATEV) load_vals();
    u32* s_vals = s_keys;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (FULL || (woff + j * WAVE) < tile_n) s_vals[pos[j]] = val[j];
    __syncthreads();
    if (LAST && a.vals_out64) {   // uniform
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const u32 p = k * BLOCK + tid;
            if (FULL || p < tile_n) { const u32 v = s_vals[p]; a.vals_out[gidx[k]] = v; a.vals_out64[gidx[k]] = (int64_t)v; }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) a.vals_out[gidx[k]] = s_vals[p];
    }
}

// Bytes in flight per CU are what this kernel lives on.  512 x 16 records with the values loaded right after the
// ranking: 104 VGPRs, two workgroups per CU (2 x 128 KB in flight), 3.80-3.86 ms per pass at N = 1e9.  Values loaded only
// before they are staged (LATEV): 80 VGPRs, 44 KB of LDS, THREE workgroups (3 x 128 KB): 3.44-3.49 ms.  512 x 24 records
// (tiles of 12288, 127 VGPRs, 60 KB of LDS, two workgroups = 2 x 192 KB, digit runs of 48 instead of 32 records and a
// third fewer look-backs): 3.29-3.31 ms against 3.56 ms on the same box -- the instantiated form.  512 x 32 spills
// (63 registers); it is not the number of waves that counts: 1024 x 8 threads x records (64 VGPRs, two workgroups of
// 16 waves) ran at the speed of 512 x 16 with two workgroups (3.76 vs 3.78 ms).
template <int BLOCK, int ITEMS, bool LAST, bool LATEV = true>
__global__ __launch_bounds__(BLOCK, (ITEMS > 16) ? 4 : (LATEV ? 6 : 4)) void seg_onesweep_kernel(SegPassArgs a) {
    constexpr int WAVES = BLOCK / WAVE;
    constexpr u32 TILE = BLOCK * ITEMS;
    __shared__ __attribute__((aligned(16))) u32 s_keys[TILE];   // reused for the values
    __shared__ u32 s_whist[WAVES * RADIX];
    __shared__ u32 s_gdelta[RADIX];
    __shared__ u32 s_wsum[RADIX / WAVE];
    __shared__ u32 s_t[RADIX + 1];
    __shared__ u32 s_b[RADIX + 1];
    __shared__ u32 s_c[NCHUNK + 1];
    __shared__ u32 s_flat;

    const int tid = threadIdx.x;
    // one tile per workgroup: the ticket of this XCD's part is requested first, the plan is read meanwhile
    u32 home = 0, t_home = 0;
    if (tid == 0) {
        home = xcc_id();
        t_home = atomicAdd(&a.ticket[home], 1u);
    }
    // the whole plan goes to LDS while the ticket is in flight: no dependent global load between the ticket and
    // the tile's first key load
    for (int i = tid; i <= RADIX; i += BLOCK) { s_t[i] = a.plan->tprefix[i]; s_b[i] = a.plan->bstart[i]; }
    if (tid <= NCHUNK) s_c[tid] = a.plan->cfirst[tid];
    for (int i = tid; i < WAVES * RADIX; i += BLOCK) s_whist[i] = 0;
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
    if (rest >= TILE)
        seg_tile<true, BLOCK, ITEMS, LAST, LATEV>(a, flat, first_flat, bucket, start, TILE, s_keys, s_whist, s_gdelta, s_wsum);
    else
        seg_tile<false, BLOCK, ITEMS, LAST, LATEV>(a, flat, first_flat, bucket, start, rest, s_keys, s_whist, s_gdelta, s_wsum);
}

}  // namespace sa
#include "radix_split.hpp"   // the three-pass plan: split pass + local finish (uses SegPlan / seg_bucket_of above)
namespace sa {

// ---- histogram of the top digit, per chunk of the input order, straight from the text -----------------------
// hist[chunk][d], d = top 8 bits of the key of every position = its first c8 = ceil(8 / b) characters.  Every
// thread owns 16 consecutive positions (one 16-byte load + the next one for the c8 - 1 characters that
// follow), rolls a c8-character window over them and counts into one of TOP_HIST_COPIES LDS histograms.
constexpr int TOP_HIST_COPIES = 8;
// LOW_K0 > 0: the histogram of the LOWEST 8 key bits instead (pass 0 of the plain LSD sort of LOW_K0-character keys,
// text_low_pass_kernel): those are the low bits of the key's last c8 characters, i.e. the same rolling window LOW_K0 - c8
// characters further on.
template <int C8>   // characters that make up the top 8 key bits: ceil(8 / b), a compile-time constant so that the window loop unrolls
__global__ __launch_bounds__(256) void top_hist_kernel(const u8* __restrict__ text, const u16* __restrict__ map, u64 n, int b,
                                                       SortGeom g, u32* __restrict__ hist, int low_k0 = 0) {
    constexpr int CS = RADIX + 1;
    constexpr u32 SPAN = 256 * 16;
    __shared__ u32 s_h[TOP_HIST_COPIES * CS];
    __shared__ u8 s_map[256];
    s_map[threadIdx.x] = (u8)map[threadIdx.x];
    for (int i = threadIdx.x; i < TOP_HIST_COPIES * CS; i += 256) s_h[i] = 0;
    __syncthreads();
    u32* my = s_h + (threadIdx.x & (TOP_HIST_COPIES - 1)) * CS;
    constexpr int c8 = C8;
    const u32 wmask = (1u << (c8 * b)) - 1u;
    const int dshift = c8 * b - 8;
    const u64 nspans = (n + SPAN - 1) / SPAN;
    const u64 per = (nspans + gridDim.x - 1) / gridDim.x;
    const u64 s_lo = (u64)blockIdx.x * per;
    const u64 s_hi = (s_lo + per < nspans) ? s_lo + per : nspans;
    auto flush = [&](u32 chunk) {
        sync_lds();
        u32 v = 0;
#pragma unroll
        for (int k = 0; k < TOP_HIST_COPIES; ++k) { v += s_h[k * CS + threadIdx.x]; s_h[k * CS + threadIdx.x] = 0; }
        if (v) atomicAdd(&hist[chunk * RADIX + threadIdx.x], v);
        sync_lds();
    };
    if (s_lo >= s_hi) return;
    u32 cur = chunk_of_tile((u32)((s_lo * SPAN) / g.tile), g.tpc);
    for (u64 sp = s_lo; sp < s_hi; ++sp) {
        const u64 base = sp * SPAN;
        const u32 c = chunk_of_tile((u32)(base / g.tile), g.tpc);   // a chunk is a whole number of tiles (8192 or 12288 records) = spans
        if (c != cur) { flush(cur); cur = c; }
        const u64 p0 = base + (u64)threadIdx.x * 16;
        if (p0 < n) {
            // 32 bytes from p0 + off (the buffer is readable TEXT_PAD >= 32 + 64 bytes past n); off = 0 for the top digit
            const u64 off = low_k0 ? (u64)(low_k0 - c8) : 0;
            uint4 x0, x1;
            __builtin_memcpy(&x0, text + p0 + off, 16);
            __builtin_memcpy(&x1, text + p0 + off + 16, 16);
            const u32 w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const u64 first = p0 + off;
            const u32 live = (first >= n) ? 0u : ((n - first < 32) ? (u32)(n - first) : 32u);   // bytes of the text among the 32
            const u32 npos = (n - p0 < 16) ? (u32)(n - p0) : 16u;                                 // positions of this thread
            u32 win = 0;
#pragma unroll
            for (int k = 0; k < 16 + c8 - 1; ++k) {
                const u32 byte = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
                const u32 code = ((u32)k < live) ? (u32)s_map[byte] : 0u;
                win = ((win << b) | code) & wmask;
                const int i = k - (c8 - 1);   // the window now ends at k: it is the one of position p0 + i
                if (i >= 0 && (u32)i < npos) atomicAdd(&my[low_k0 ? (win & 255u) : (win >> dshift)], 1u);
            }
        }
    }
    flush(cur);
}

// ---- top-digit pass straight from the text -------------------------------------------------------------
// The u64 keys of the top-digit pass exist only to be read once: this kernel assembles them in registers
// from the text instead (no key array, no key-generation kernel: 1 byte read + 8 bytes written per record
// instead of 9 + 16).  A tile's text bytes (+ a halo of k0 - 1) are staged in LDS as alphabet codes;
// every lane builds the keys of its 16 wave-striped positions character by character (one LDS byte read
// and one shift-or per character and key), ranks them by the top digit and the pass proceeds as in
// radix_onesweep_kernel with iota values; the top digit of a sorted slot travels through LDS beside the
// narrow key (the narrow key does not contain it).
struct TextPassArgs {
    const u8* text;
    const u16* map;         // CodeMap::code in device memory
    u64 n;
    int b, k0, begin_bit;   // key = k0 characters of b bits, MSB first, in bits [begin_bit, 64); b <= 8
    u32* keys_out32;        // (u32)(key >> begin_bit); EXT: (u32)(key >> (begin_bit + 16))
    u16* ext_out16;         // EXT (48-bit remainder, radix_narrow48.hpp): (u16)(key >> begin_bit)
    u32* vals_out;          // text position
    SortGeom g;
    const u32* digit_base;  // [NCHUNK][RADIX] of the top digit
    u64* status;
    u32* ticket;
    u32 epoch;
    DeviceStatus* dstat;
    u32 incl_mask;
    u32* cursor;            // CLAIM form: records of a digit placed so far (zeroed by the host): cursor[digit * cursor_stride]
    u32 cursor_stride;      // 64 words by default: every digit's counter in a 256-byte line of its own (SA_HIP_CURSOR_PAD=0: 1)
};
constexpr int TEXT_HALO = 64;   // >= k0 - 1 (k0 * b <= 40)
// positions per thread of the text-sourced top-digit pass.  24 (tiles of 12288 as in the narrow passes: 118 VGPRs, 71 KB of
// LDS, two workgroups per CU) measured 3.67 against 3.61 ms for 16 (three workgroups) on the same box: unlike the narrow
// passes this one does not live on bytes in flight, it is bound by its per-record work.
#ifndef SA_TEXT_ITEMS
#define SA_TEXT_ITEMS 16
#endif
constexpr int TEXT_ITEMS = SA_TEXT_ITEMS;
constexpr u32 TEXT_TILE = 512u * TEXT_ITEMS;

// acc |= x << s (s uniform), accumulator updated in place
__device__ __forceinline__ void shl_or_inplace(u32& acc, u32 x, int s) {
    asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(s));
}

// (b and k0 stay run-time values: with b = 5, k0 = 8 as compile-time constants the compiler merges a key's eight byte
// reads into one unaligned ds_read_b64 and the pass gets slower, 4.54 vs 3.69 ms at N = 1e9.)
// CLAIM (round 4): the pass need not be stable when the three-pass plan follows (radix_split.hpp: its local pass orders by
// (key, suffix) whatever order it finds): a tile then claims its place inside a digit with one returning global atomic per
// non-empty digit instead of publishing its counts and looking back over its predecessors, as seg_split_kernel does.
// RANKA (with CLAIM): the place inside the tile's digit from one returning LDS atomic per record on ONE tile-wide counter array
// instead of the ballot-match masks (about 40 vector instructions per record less; not stable).
template <bool FULL, int BLOCK, bool EXT, bool CLAIM = false, bool RANKA = false>
__device__ __forceinline__ void text_top_tile(const TextPassArgs& a, const u32 tile, const u32 chunk,
                                              const u32 tile_n, u32* s_keys, u32* s_whist, u32* s_gdelta, u32* s_wsum,
                                              u8* s_code, const u8* s_map, u16* s_ext) {
    constexpr int WAVES = BLOCK / WAVE;
    constexpr int ITEMS = TEXT_ITEMS;
    constexpr u32 TILE = BLOCK * ITEMS;
    static_assert(TILE % 16 == 0, "16-byte text loads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 first_tile = chunk * a.g.tpc;
    const u64 tile_base = (u64)tile * TILE;
    const u32 woff = (u32)wave * (WAVE * ITEMS) + lane;

    const u32 dbase = (tid < RADIX) ? a.digit_base[(CLAIM ? 0u : chunk) * RADIX + tid] : 0u;   // needed after the look-back (CLAIM: the digit's start)

    // 0. text -> codes in LDS: 16 bytes per thread, the halo by the first TEXT_HALO / 16 threads
    //    (the text buffer is readable for TEXT_PAD >= 16 bytes past n; a load starts below n or is skipped)
    auto stage16 = [&](u32 local, auto checked) {
        constexpr bool CHECK = decltype(checked)::value;   // false: all 16 positions are known to lie below n
        const u64 p0 = tile_base + local;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (!CHECK || p0 < a.n) x = *reinterpret_cast<const uint4*>(a.text + p0);
        const u32 w[4] = {x.x, x.y, x.z, x.w};
        const u32 live = CHECK ? (u32)((a.n > p0) ? ((a.n - p0 < 16) ? (a.n - p0) : 16) : 0) : 16u;
        u32 o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const u32 byte = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
            const u32 code = (!CHECK || (u32)k < live) ? (u32)s_map[byte] : 0u;
            o[k >> 2] |= code << ((k & 3) * 8);
        }
        *reinterpret_cast<uint4*>(s_code + local) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    for (u32 local = (u32)tid * 16u; local < TILE; local += BLOCK * 16u) {
        if (FULL) stage16(local, std::false_type{}); else stage16(local, std::true_type{});
    }
    if (tid < TEXT_HALO / 16) stage16(TILE + (u32)tid * 16u, std::true_type{});
    __syncthreads();

    // 1. keys of the lane's positions woff + 64 j, one character per step: hi:lo = bits [32,64):[0,32)
    u32 hi[ITEMS], lo[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { hi[j] = 0; lo[j] = 0; }
    // three loops with one body each: characters entirely in the upper word, at most one character that straddles
    // bit 32, characters entirely in the lower word (the shift amounts are scalar, the accumulate is one in-place
    // v_lshl_or_b32 per character and key)
    const u8* cp = s_code + woff;
    {
        int c = 0;
        for (; c < a.k0 && 64 - a.b * (c + 1) >= 32; ++c, ++cp) {
            const int s_hi = 32 - a.b * (c + 1);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) shl_or_inplace(hi[j], (u32)cp[j * WAVE], s_hi);
        }
        if (c < a.k0 && 64 - a.b * c > 32) {
            const int sh = 64 - a.b * (c + 1);   // 0 < sh < 32 < sh + b
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 code = cp[j * WAVE];
                hi[j] |= code >> (32 - sh);
                lo[j] |= code << sh;
            }
            ++c; ++cp;
        }
        for (; c < a.k0; ++c, ++cp) {
            const int sh = 64 - a.b * (c + 1);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) shl_or_inplace(lo[j], (u32)cp[j * WAVE], sh);
        }
    }
    u32 key[ITEMS];   // narrow keys: bits [begin_bit, begin_bit + 32) of hi:lo (EXT: bits [begin_bit + 16, begin_bit + 48))
    u32 ext[EXT ? ITEMS : 1];   // EXT: bits [begin_bit, begin_bit + 16)
    const int kb0 = EXT ? a.begin_bit + 16 : a.begin_bit;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        key[j] = (kb0 >= 32) ? (hi[j] >> (kb0 - 32)) : __builtin_amdgcn_alignbit(hi[j], lo[j], (u32)kb0);
        if (EXT) ext[j] = __builtin_amdgcn_alignbit(hi[j], lo[j], (u32)a.begin_bit) & 0xFFFFu;   // begin_bit < 24
    }

    // 2. rank by the top digit (bits 24..31 of hi)
    u32 rd[ITEMS];
    u32* wh = s_whist + (RANKA ? 0 : wave * RADIX);
    if constexpr (RANKA) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 d = hi[j] >> 24;
            const u32 r = atomicAdd(&s_whist[(FULL || (woff + j * WAVE) < tile_n) ? d : (u32)RADIX], 1u);   // (a slot beyond the text: the spare word of wave 1's counters)
            rd[j] = r | (d << 16);
        }
        sync_lds();
    } else {
        wave_rank<FULL>(hi, 24, 255u, woff, tile_n, wh, rd);
        __syncthreads();   // also: every read of s_code is done (it becomes the digit array below)
    }

    // 3. tile digit counts -> aggregate -> exclusive scan over digits
    u32 count = 0, excl = 0, claim = 0;
    if (tid < RADIX) {
        u32 c = 0;
        if constexpr (RANKA) { c = s_whist[tid]; }
        else {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const u32 t = s_whist[w * RADIX + tid];
            s_whist[w * RADIX + tid] = c;
            c += t;
        }
        }
        count = c;
        if (CLAIM) { if (c) claim = atomicAdd(&a.cursor[(size_t)tid * a.cursor_stride], c); }   // requested now, needed for the stores
        else
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
        if constexpr (RANKA) s_whist[tid] = excl;
        else {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += excl;
        }
    }
    __syncthreads();

    // 4. narrow keys and their top digits -> LDS at the tile-local sorted position
    u8* s_dig = s_code;
    u32 pos[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 d = rd[j] >> 16;
        pos[j] = wh[d] + (rd[j] & 0xFFFFu);
        if (FULL || (woff + j * WAVE) < tile_n) {
            s_keys[pos[j]] = key[j]; s_dig[pos[j]] = (u8)d;
            if (EXT) s_ext[pos[j]] = (u16)ext[j];   // the low 16 key bits beside the upper 32: one staging round for both
        }
    }
    __syncthreads();

    // 5. look-back
    if (tid < RADIX) {
        u32 prefix = claim;
        if (!CLAIM && tile > first_tile) {
            prefix = lookback_prefix(a.status, tile, first_tile, (u32)tid, a.epoch, a.dstat);
            if (((tile - first_tile) & a.incl_mask) == a.incl_mask)
                __hip_atomic_store(&a.status[(u64)tile * RADIX + tid], pack_status(a.epoch, FLAG_INCL, prefix + count),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gdelta[tid] = dbase + prefix - excl;
    }
    __syncthreads();

    // 6. coalesced stores per digit run: keys, then the positions
    u32 gidx[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) {
            gidx[k] = s_gdelta[s_dig[p]] + p;
            a.keys_out32[gidx[k]] = s_keys[p];
            if (EXT) a.ext_out16[gidx[k]] = s_ext[p];
        }
    }
    __syncthreads();
    u32* s_vals = s_keys;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (FULL || (woff + j * WAVE) < tile_n) s_vals[pos[j]] = (u32)tile_base + woff + j * WAVE;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) a.vals_out[gidx[k]] = s_vals[p];
    }
}

// 79 VGPRs and 50 KB of LDS: three workgroups (24 waves) per CU
template <int BLOCK, bool EXT = false, bool CLAIM = false, bool RANKA = false>
__global__ __launch_bounds__(BLOCK, (TEXT_ITEMS > 16 || EXT) ? 4 : 6) void text_top_pass_kernel(TextPassArgs a) {
    constexpr int WAVES = BLOCK / WAVE;
    constexpr u32 TILE = BLOCK * TEXT_ITEMS;
    __shared__ __attribute__((aligned(16))) u32 s_keys[TILE];
    __shared__ u32 s_whist[WAVES * RADIX];
    __shared__ u32 s_gdelta[RADIX];
    __shared__ u32 s_wsum[RADIX / WAVE];
    __shared__ __attribute__((aligned(16))) u8 s_code[TILE + TEXT_HALO];   // codes, later the top digit per sorted slot
    __shared__ u16 s_ext[EXT ? TILE : 2];
    __shared__ u8 s_map[256];
    __shared__ u32 s_tile;
    __shared__ u32 s_chunk;

    const int tid = threadIdx.x;
    if (tid == 0) {
        u32 tile = 0xFFFFFFFFu, chunk = 0;
        const u32 home = xcc_id();
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
    if (tid < 256) s_map[tid] = (u8)a.map[tid];
    for (int i = tid; i < WAVES * RADIX; i += BLOCK) s_whist[i] = 0;
    __syncthreads();
    const u32 tile = s_tile;
    if (tile == 0xFFFFFFFFu) return;
    const u32 chunk = s_chunk;
    const u64 rest = a.n - (u64)tile * TILE;
    if (rest >= (u64)TILE)
        text_top_tile<true, BLOCK, EXT, CLAIM, RANKA>(a, tile, chunk, TILE, s_keys, s_whist, s_gdelta, s_wsum, s_code, s_map, s_ext);
    else
        text_top_tile<false, BLOCK, EXT, CLAIM, RANKA>(a, tile, chunk, (u32)rest, s_keys, s_whist, s_gdelta, s_wsum, s_code, s_map, s_ext);
}

// ---- pass 0 of the plain LSD sort straight from the text ---------------------------------------------------
// Keys of more than 40 bits (word / name / log-like text: 12 characters of 5 bits) are sorted as (u64 key, u32 suffix)
// records by radix_sort_pairs.  Its pass 0 used to read a key array that keygen_kernel had written for that one read
// (9 + 20 bytes per character); this kernel assembles the keys of a tile in registers from the text as
// text_top_pass_kernel does and then IS pass 0: ranked by the LOWEST digit, values = positions, u64 keys + u32 values
// out, the (chunk, next digit) histogram of pass 1 counted on the way: 1 + 12 bytes per character.
struct TextLowArgs {
    SortPassArgs p;         // keys_in / vals_in unused; keys_out u64, vals_out, geometry, digits, look-back state
    const u8* text;
    const u16* map;
    u64 n;
    int b, k0;              // key = k0 characters of b bits, MSB first, in bits [64 - b * k0, 64); b <= 8
};

template <bool FULL, int BLOCK>
__device__ __forceinline__ void text_low_tile(const TextLowArgs& t, const u32 tile, const u32 chunk, const u32 tile_n,
                                              u64* s_keys, u32* s_whist, uint2* s_tab, u32* s_wsum, const u8* s_map) {
    constexpr int WAVES = BLOCK / WAVE;
    constexpr int ITEMS = SORT_ITEMS;
    constexpr u32 TILE = BLOCK * ITEMS;
    const SortPassArgs& a = t.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 first_tile = chunk * a.g.tpc;
    const u64 tile_base = (u64)tile * TILE;
    const u32 woff = (u32)wave * (WAVE * ITEMS) + lane;
    const u32 dbase = (tid < RADIX) ? a.digit_base[chunk * RADIX + tid] : 0u;

    // 0. text -> codes, staged in the LDS region that holds the sorted keys later (every code is read before the first
    //    key is written: two barriers lie between)
    u8* s_code = reinterpret_cast<u8*>(s_keys);
    auto stage16 = [&](u32 local, auto checked) {
        constexpr bool CHECK = decltype(checked)::value;
        const u64 p0 = tile_base + local;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (!CHECK || p0 < t.n) x = *reinterpret_cast<const uint4*>(t.text + p0);
        const u32 w[4] = {x.x, x.y, x.z, x.w};
        const u32 live = CHECK ? (u32)((t.n > p0) ? ((t.n - p0 < 16) ? (t.n - p0) : 16) : 0) : 16u;
        u32 o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const u32 byte = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
            const u32 code = (!CHECK || (u32)k < live) ? (u32)s_map[byte] : 0u;
            o[k >> 2] |= code << ((k & 3) * 8);
        }
        *reinterpret_cast<uint4*>(s_code + local) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    for (u32 local = (u32)tid * 16u; local < TILE; local += BLOCK * 16u) {
        if (FULL) stage16(local, std::false_type{}); else stage16(local, std::true_type{});
    }
    if (tid < TEXT_HALO / 16) stage16(TILE + (u32)tid * 16u, std::true_type{});
    __syncthreads();

    // 1. keys of the lane's positions woff + 64 j, one character per step (see text_top_tile)
    u32 hi[ITEMS], lo[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { hi[j] = 0; lo[j] = 0; }
    const u8* cp = s_code + woff;
    {
        int c = 0;
        for (; c < t.k0 && 64 - t.b * (c + 1) >= 32; ++c, ++cp) {
            const int s_hi = 32 - t.b * (c + 1);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) shl_or_inplace(hi[j], (u32)cp[j * WAVE], s_hi);
        }
        if (c < t.k0 && 64 - t.b * c > 32) {
            const int sh = 64 - t.b * (c + 1);   // 0 < sh < 32 < sh + b
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 code = cp[j * WAVE];
                hi[j] |= code >> (32 - sh);
                lo[j] |= code << sh;
            }
            ++c; ++cp;
        }
        for (; c < t.k0; ++c, ++cp) {
            const int sh = 64 - t.b * (c + 1);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) shl_or_inplace(lo[j], (u32)cp[j * WAVE], sh);
        }
    }
    u64 key[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) key[j] = ((u64)hi[j] << 32) | lo[j];

    // 2. rank by the pass's digit
    u32 rd[ITEMS];
    u32* wh = s_whist + wave * RADIX;
    wave_rank<FULL>(key, a.shift, a.mask, woff, tile_n, wh, rd);
    __syncthreads();   // also: every read of the staged codes is done

    // 3. tile digit counts -> aggregate -> exclusive scan over digits
    u32 count = 0, excl = 0;
    if (tid < RADIX) {
        u32 c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const u32 x = s_whist[w * RADIX + tid];
            s_whist[w * RADIX + tid] = c;
            c += x;
        }
        count = c;
        __hip_atomic_store(&a.status[(u64)tile * RADIX + tid], pack_status(a.epoch, tile == first_tile ? FLAG_INCL : FLAG_AGG, count),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32 incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 x = __shfl_up(incl, o);
            if (lane >= o) incl += x;
        }
        if (lane == 63) s_wsum[wave] = incl;
        excl = incl - c;
    }
    __syncthreads();
    if (tid < RADIX) {
        for (int i = 0; i < wave; ++i) excl += s_wsum[i];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += excl;
    }
    __syncthreads();

    // 4. keys -> LDS at their tile-local sorted position
    u32 pos[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        pos[j] = wh[rd[j] >> 16] + (rd[j] & 0xFFFFu);
        if (FULL || (woff + j * WAVE) < tile_n) s_keys[pos[j]] = key[j];
    }
    __syncthreads();

    // 5. look-back; the other lanes clear the (chunk, next digit) histogram that reuses s_whist
    const bool has_next = a.next_shift >= 0;
    if (has_next) for (int i = tid; i < NCHUNK * RADIX; i += BLOCK) s_whist[i] = 0;
    if (tid < RADIX) {
        u32 prefix = 0;
        if (tile > first_tile) {
            prefix = lookback_prefix(a.status, tile, first_tile, (u32)tid, a.epoch, a.dstat);
            if (((tile - first_tile) & a.incl_mask) == a.incl_mask)
                __hip_atomic_store(&a.status[(u64)tile * RADIX + tid], pack_status(a.epoch, FLAG_INCL, prefix + count),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const u32 g0 = dbase + prefix;
        const u32 gdelta = g0 - excl;
        const u32 c0 = chunk_of_tile(g0 >> a.g.tile_shift, a.g.tpc);
        const u64 bnd = (u64)(c0 + 1) * a.g.tpc << a.g.tile_shift;
        u32 thr = 0xFFFFu;
        if (c0 + 1 < (u32)NCHUNK && bnd < (u64)g0 + count) thr = (u32)(bnd - gdelta);
        s_tab[tid] = make_uint2(gdelta, (c0 << 16) | thr);
    }
    __syncthreads();

    // 6. coalesced stores per digit run (+ pass 1's per-chunk histogram), then the positions
    u32 gidx[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) {
            const u64 kk = s_keys[p];
            const uint2 tb = s_tab[digit_of(kk, a.shift, a.mask)];
            gidx[k] = tb.x + p;
            a.keys_out[gidx[k]] = kk;
            if (has_next) {
                const u32 dn = digit_of(kk, a.next_shift, a.next_mask);
                const u32 cn = (tb.y >> 16) + (p >= (tb.y & 0xFFFFu) ? 1u : 0u);
                atomicAdd(&s_whist[cn * RADIX + dn], 1u);
            }
        }
    }
    sync_lds();
    if (has_next) {
        for (int i = tid; i < NCHUNK * RADIX; i += BLOCK) {
            const u32 v = s_whist[i];
            if (v) atomicAdd(&a.next_hist[i], v);
        }
    }
    u32* s_vals = reinterpret_cast<u32*>(s_keys);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (FULL || (woff + j * WAVE) < tile_n) s_vals[pos[j]] = (u32)tile_base + woff + j * WAVE;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) a.vals_out[gidx[k]] = s_vals[p];
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK, 4) void text_low_pass_kernel(TextLowArgs t) {
    constexpr int WAVES = BLOCK / WAVE;
    constexpr u32 TILE = BLOCK * SORT_ITEMS;
    constexpr int WH = (WAVES * RADIX > NCHUNK * RADIX) ? WAVES * RADIX : NCHUNK * RADIX;
    static_assert(TILE * 8 >= TILE + TEXT_HALO, "the staged codes fit the key array's LDS");
    __shared__ __attribute__((aligned(16))) u64 s_keys[TILE];
    __shared__ u32 s_whist[WH];
    __shared__ uint2 s_tab[RADIX];
    __shared__ u32 s_wsum[RADIX / WAVE];
    __shared__ u8 s_map[256];
    __shared__ u32 s_tile;
    __shared__ u32 s_chunk;
    const SortPassArgs& a = t.p;
    const int tid = threadIdx.x;
    if (tid == 0) {
        u32 tile = 0xFFFFFFFFu, chunk = 0;
        const u32 home = xcc_id();
        for (int k = 0; k < NCHUNK; ++k) {
            const u32 c = (home + k) & (NCHUNK - 1);
            const u32 first = c * a.g.tpc;
            if (first >= a.g.tiles) continue;
            const u32 cnt = (a.g.tiles - first) < a.g.tpc ? (a.g.tiles - first) : a.g.tpc;
            const u32 x = atomicAdd(&a.ticket[c], 1u);
            if (x < cnt) { tile = first + x; chunk = c; break; }
        }
        s_tile = tile;
        s_chunk = chunk;
    }
    if (tid < 256) s_map[tid] = (u8)t.map[tid];
    for (int i = tid; i < WAVES * RADIX; i += BLOCK) s_whist[i] = 0;
    __syncthreads();
    const u32 tile = s_tile;
    if (tile == 0xFFFFFFFFu) return;
    const u32 chunk = s_chunk;
    const u64 rest = t.n - (u64)tile * TILE;
    if (rest >= (u64)TILE)
        text_low_tile<true, BLOCK>(t, tile, chunk, TILE, s_keys, s_whist, s_tab, s_wsum, s_map);
    else
        text_low_tile<false, BLOCK>(t, tile, chunk, (u32)rest, s_keys, s_whist, s_tab, s_wsum, s_map);
}

// ---- host driver --------------------------------------------------------------------------------------
struct NarrowWorkspace {
    SegPlan* plan = nullptr;
    u32* hist = nullptr;     // [NARROW_MAX_PASSES][RADIX][RADIX]
    u32* base = nullptr;     // [RADIX][RADIX]
    u32* tickets = nullptr;  // [NARROW_MAX_PASSES][NCHUNK]
    u16* map_dev = nullptr;  // CodeMap of the text pass
    CodeMap map_host;        // source of the asynchronous copy (must outlive the call)
    // the three-pass plan (radix_split.hpp)
    bool split_enabled = true;     // SA_HIP_SPLIT=0: always the LSD passes
    bool split_used = false;       // of the last sort
    u32 split_max_seen = 0;        // largest sub-bucket of the last sort that looked (0: the plan was not considered)
    u32* split_hist = nullptr;     // [RADIX][SPLIT_NB]
    u32* split_base = nullptr;     // [RADIX][SPLIT_NB]
    u32* split_cursor = nullptr;   // [RADIX][SPLIT_NB]: the ATOMIC form's claims
    u32* top_cursor = nullptr;     // [RADIX]: the claims of the top-digit pass's CLAIM form
    bool cursor_pad = true;        // SA_HIP_CURSOR_PAD=0: the 256 claim counters of the top-digit pass packed into 1 KB (measured on a slow host: 4.1-4.2 against 3.5-3.8 ms)
    bool split_cursor_t = false;   // SA_HIP_SPLIT_CURSOR_T=1: the split pass's counters as [bin][bucket] (a tile's claims 1 KB apart; measured no better)
    bool top_atomic_ranks = false; // SA_HIP_TOP_ARANKS=1: ... with LDS-atomic ranks as well
    bool top_claims = true;        // SA_HIP_TOP_CLAIMS=0: the top-digit pass always in its stable form (published counts + look-back)
    bool split_atomic = true;      // SA_HIP_SPLIT_ATOMIC=0: the split pass with published counts and a look-back per bucket instead of claims by global atomics
                                   // (measured: 4.25 against 3.05-3.45 ms at n = 1e9, profiles/r04_split_plan_atomic_ab.log)
    u32* split_sub = nullptr;      // [(RADIX << SPLIT_BITS) + 1] sub-bucket starts | [16] largest group per level
    u64* split_status = nullptr;   // [split_tiles][SPLIT_NB], allocated with the first sort that takes the plan
    u32 split_tiles = 0;
    u32 split_epoch = 0;           // epoch of the last split pass (the granules are re-zeroed when the sort's epoch has wrapped)
    u32* host_word = nullptr;      // pinned, 16 words: the largest group per level comes here
    static size_t split_table_bytes() { return (size_t)RADIX * SPLIT_NB * sizeof(u32); }
    static size_t split_sub_words() { return ((size_t)RADIX << SPLIT_BITS) + 1; }
    u32* split_levels_dev() const { return split_sub + split_sub_words(); }   // [16]: largest group per level
    u32 split_cap = LOCAL_CAP;     // SA_HIP_SPLIT_CAP (tests: a smaller bound makes small texts take more levels)
    int split_rb = 0;              // level of the last sort that took the plan
    int local_bin_bits = 12;       // SA_HIP_LOCAL_BINS=11: 2048 bins in the local pass
    bool split_flags = true;       // SA_HIP_SPLIT_FLAGS=0: the first flags pass stays a pass of its own
    bool local_big = false;        // SA_HIP_LOCAL_BIG=1: always the large form of the local pass (tests, A/B)
    bool split_big = false;        // of the last sort: the local pass ran in its large form
    bool split_flags_done = false; // of the last sort: the local pass has written the directory and staged the active records
    int split_items = 28;          // SA_HIP_SPLIT_ITEMS=24 / 28 / 32: tiles of 12288 / 14336 / 16384 records in the split pass (32 spills 33 registers)
    static size_t hist_bytes() { return (size_t)NARROW_MAX_PASSES * RADIX * RADIX * sizeof(u32); }
    int init() {
        SA_HIP_CHECK(hipMalloc(&plan, sizeof(SegPlan)));
        SA_HIP_CHECK(hipMalloc(&hist, hist_bytes()));
        SA_HIP_CHECK(hipMalloc(&base, (size_t)RADIX * RADIX * sizeof(u32)));
        SA_HIP_CHECK(hipMalloc(&tickets, (size_t)NARROW_MAX_PASSES * NCHUNK * sizeof(u32)));
        SA_HIP_CHECK(hipMalloc(&map_dev, sizeof(CodeMap)));
        SA_HIP_CHECK(hipMalloc(&split_hist, split_table_bytes()));
        SA_HIP_CHECK(hipMalloc(&split_base, split_table_bytes()));
        SA_HIP_CHECK(hipMalloc(&split_cursor, split_table_bytes()));
        SA_HIP_CHECK(hipMalloc(&top_cursor, (size_t)RADIX * 64 * sizeof(u32)));
        SA_HIP_CHECK(hipMalloc(&split_sub, (split_sub_words() + 16) * sizeof(u32)));
        SA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&host_word), 64, hipHostMallocDefault));
        if (const char* e = diag_env("SA_HIP_SPLIT")) split_enabled = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_CURSOR_PAD")) cursor_pad = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_SPLIT_CURSOR_T")) split_cursor_t = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_TOP_CLAIMS")) top_claims = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_TOP_ARANKS")) top_atomic_ranks = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_LOCAL_BIG")) local_big = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_SPLIT_ATOMIC")) split_atomic = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_SPLIT_FLAGS")) split_flags = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_LOCAL_BINS")) local_bin_bits = (atoi(e) == 11) ? 11 : 12;
        if (const char* e = diag_env("SA_HIP_SPLIT_ITEMS")) { const int v = atoi(e); split_items = (v == 24 || v == 28) ? v : 32; }
        if (const char* e = diag_env("SA_HIP_SPLIT_CAP")) { const int v = atoi(e); if (v >= 64 && v <= (int)LOCAL_CAP) split_cap = (u32)v; }
        return 0;
    }
    // (zeroed ON THE SORT'S STREAM: a hipMemset on the null stream is not ordered against a non-blocking stream and was still
    //  clearing granules while the split pass published them)
    int ensure_split_status(u32 tiles, hipStream_t stream) {
        if (tiles <= split_tiles) return 0;
        if (split_status) { SA_HIP_CHECK(hipStreamSynchronize(stream)); (void)hipFree(split_status); }
        split_status = nullptr; split_tiles = 0;
        SA_HIP_CHECK(hipMalloc(&split_status, (size_t)tiles * SPLIT_NB * sizeof(u64)));
        SA_HIP_CHECK(hipMemsetAsync(split_status, 0, (size_t)tiles * SPLIT_NB * sizeof(u64), stream));
        split_tiles = tiles;
        return 0;
    }
    void destroy() {
        if (plan) (void)hipFree(plan);
        if (hist) (void)hipFree(hist);
        if (base) (void)hipFree(base);
        if (tickets) (void)hipFree(tickets);
        if (map_dev) (void)hipFree(map_dev);
        if (split_hist) (void)hipFree(split_hist);
        if (split_base) (void)hipFree(split_base);
        if (split_cursor) (void)hipFree(split_cursor);
        if (top_cursor) (void)hipFree(top_cursor);
        if (split_sub) (void)hipFree(split_sub);
        if (split_status) (void)hipFree(split_status);
        if (host_word) (void)hipHostFree(host_word);
        map_dev = nullptr;
        plan = nullptr; hist = nullptr; base = nullptr; tickets = nullptr;
        split_hist = split_base = split_sub = split_cursor = top_cursor = nullptr; split_status = nullptr; split_tiles = 0; host_word = nullptr;
    }
};

// Keys of at most 40 bits and a tile size of 8192; below ~4M records the flat tile count (n / tile + 256)
// would not fit the status array sized for the plain sort, and nothing is to be gained there anyway.
inline bool narrow_sort_applies(const RadixWorkspace& ws, u64 n, int begin_bit) {
    return ws.block == 512 && begin_bit >= 24 && begin_bit < 56 && n >= (1u << 22) &&
           n / ws.tile() + RADIX + 1 <= ws.max_tiles;
}

// Where the keys come from when the top-digit pass reads the text itself (text_top_pass_kernel)
struct TextSource {
    const u8* text;
    int b, k0;
};
inline bool text_pass_applies(int b, int k0) { return b <= 8 && k0 - 1 <= TEXT_HALO; }

// Code map to the device + per-chunk histogram of the top digit of every position's key into ws.hist(0)
// (radix_prepare() before): what a text-sourced narrow sort needs in place of key generation.
inline int narrow_text_histogram(RadixWorkspace& ws, NarrowWorkspace& nw, hipStream_t stream, const u8* text, const CodeMap& map,
                                 u32 n, int b) {
    nw.map_host = map;
    SA_HIP_CHECK(hipMemcpyAsync(nw.map_dev, nw.map_host.code, sizeof(CodeMap), hipMemcpyHostToDevice, stream));
    const SortGeom g = make_geom(n, TEXT_TILE);   // the chunks of text_top_pass_kernel (its tile is not a power of two: tile_shift unused)
    const u32 spans = div_up(n, 4096);
    const dim3 grid(spans < 2048u ? spans : 2048u), block(256);
    switch ((8 + b - 1) / b) {
        case 1: hipLaunchKernelGGL(top_hist_kernel<1>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0)); break;
        case 2: hipLaunchKernelGGL(top_hist_kernel<2>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0)); break;
        case 3: hipLaunchKernelGGL(top_hist_kernel<3>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0)); break;
        case 4: hipLaunchKernelGGL(top_hist_kernel<4>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0)); break;
        default: hipLaunchKernelGGL(top_hist_kernel<8>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0)); break;
    }
    return 0;
}

// What a text-sourced pass 0 of the plain sort needs in place of key generation: the code map on the device and the
// per-chunk histogram of the LOWEST digit of every position's key in ws.hist(0) (radix_prepare() before).
inline int wide_text_histogram(RadixWorkspace& ws, NarrowWorkspace& nw, hipStream_t stream, const u8* text, const CodeMap& map,
                               u32 n, int b, int k0) {
    nw.map_host = map;
    SA_HIP_CHECK(hipMemcpyAsync(nw.map_dev, nw.map_host.code, sizeof(CodeMap), hipMemcpyHostToDevice, stream));
    const SortGeom g = make_geom(n, ws.tile());
    const u32 spans = div_up(n, 4096);
    const dim3 grid(spans < 2048u ? spans : 2048u), block(256);
    switch ((8 + b - 1) / b) {
        case 1: hipLaunchKernelGGL(top_hist_kernel<1>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0), k0); break;
        case 2: hipLaunchKernelGGL(top_hist_kernel<2>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0), k0); break;
        case 3: hipLaunchKernelGGL(top_hist_kernel<3>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0), k0); break;
        case 4: hipLaunchKernelGGL(top_hist_kernel<4>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0), k0); break;
        default: hipLaunchKernelGGL(top_hist_kernel<8>, grid, block, 0, stream, text, nw.map_dev, (u64)n, b, g, ws.hist(0), k0); break;
    }
    return 0;
}
struct WideTextCtx {
    const u8* text;
    const u16* map;
    u64 n;
    int b, k0;
};
inline void launch_text_low_pass(void* ctx, hipStream_t stream, const SortPassArgs& a, u32 grid) {
    const WideTextCtx* c = static_cast<const WideTextCtx*>(ctx);
    TextLowArgs t;
    t.p = a; t.text = c->text; t.map = c->map; t.n = c->n; t.b = c->b; t.k0 = c->k0;
    hipLaunchKernelGGL((text_low_pass_kernel<512>), dim3(grid), dim3(512), 0, stream, t);
}

// Sort n records (key[i], i) by key bits [begin_bit, 64), stable.
//   src == nullptr: keysA holds the u64 keys and the per-chunk histogram of their top digit has been accumulated
//                   into ws.hist(0) by the producer (radix_prepare() before, keygen_kernel);
//   src != nullptr: keysA holds nothing yet, the keys are the first src->k0 characters of every text position
//                   (radix_prepare() + narrow_text_histogram() before).
// keysA / keysB and valsA / valsB are the ping-pong buffers of the plain sort (n * 8 and n * 4 bytes);
// narrow keys use the first n * 4 bytes of a key buffer.  Result: *keys_res (u64, full keys), *vals_res.
// keep_narrow: the last pass does not rebuild the u64 keys; *keys_res then points at u32[n] narrow keys, the full key of
// slot j being (bucket(j) << 56) | (narrow[j] << begin_bit) with the bucket bounds in nw.plan->bstart (8 bytes per
// record less written here, 4 less read by whoever consumes the keys).
inline int radix_sort_narrow(RadixWorkspace& ws, NarrowWorkspace& nw, hipStream_t stream, u64* keysA, u32* valsA, u64* keysB,
                             u32* valsB, u32 n, int begin_bit, u64** keys_res, u32** vals_res, const TextSource* src = nullptr,
                             bool keep_narrow = false, int64_t* vals_res64 = nullptr, const LocalFlagsRequest* flags_req = nullptr) {
    int rc;
    const int lo_bits = 56 - begin_bit;                       // 1 .. 32
    const int np = (lo_bits + RADIX_BITS - 1) / RADIX_BITS;   // narrow passes, 1 .. 4
    const SortGeom g = make_geom(n, src ? TEXT_TILE : ws.tile());
    SA_HIP_CHECK(hipMemsetAsync(nw.tickets, 0, (size_t)NARROW_MAX_PASSES * NCHUNK * sizeof(u32), stream));

    // the three-pass plan (radix_split.hpp) is considered for this sort; its passes do not need the top-digit pass to be stable,
    // so that pass runs in its CLAIM form first -- and once more, stable, should the plan be declined (skewed text)
    const bool try_split = nw.split_enabled && keep_narrow && ws.block == 512 && lo_bits - LOCAL_BIN_BITS >= 1;
    const bool optimistic = try_split && src && nw.top_claims;
    auto launch_text_top = [&](bool claim) -> int {
        TextPassArgs t;
        t.text = src->text; t.map = nw.map_dev; t.n = n; t.b = src->b; t.k0 = src->k0; t.begin_bit = begin_bit;
        t.keys_out32 = reinterpret_cast<u32*>(keysB); t.ext_out16 = nullptr; t.vals_out = valsB; t.g = g; t.digit_base = ws.base();
        t.status = ws.status; t.ticket = ws.tickets(); t.epoch = ws.epoch; t.dstat = ws.dstat; t.incl_mask = SA_INCL_MASK;
        t.cursor = nw.top_cursor; t.cursor_stride = nw.cursor_pad ? 64u : 1u;
        int r;
        if (claim) SA_HIP_CHECK(hipMemsetAsync(nw.top_cursor, 0, (size_t)RADIX * 64 * sizeof(u32), stream));
        if ((r = ws.timer.start(stream, 1))) return r;
        if (claim && nw.top_atomic_ranks) hipLaunchKernelGGL((text_top_pass_kernel<512, false, true, true>), dim3(g.tiles), dim3(512), 0, stream, t);
        else if (claim) hipLaunchKernelGGL((text_top_pass_kernel<512, false, true>), dim3(g.tiles), dim3(512), 0, stream, t);
        else hipLaunchKernelGGL((text_top_pass_kernel<512>), dim3(g.tiles), dim3(512), 0, stream, t);
        if ((r = ws.timer.stop(stream, (u64)n * 9u))) return r;
        ws.pass_records += n; ws.pass_bytes += (u64)n * 9u; ws.passes += 1;
        return 0;
    };
    // top digit: values generated, keys leave as their low bits -- from the text, or from the u64 key array by the
    // ordinary one-sweep pass
    {
        if (++ws.epoch >= (1u << 30)) {
            SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
            ws.epoch = 1;
        }
        hipLaunchKernelGGL(radix_scan_hist_kernel, dim3(1), dim3(256), 0, stream, ws.hist(0), ws.base());
        if (src) {
            if ((rc = launch_text_top(optimistic))) return rc;
        } else {
            SortPassArgs a;
            a.keys_in = keysA; a.vals_in = nullptr; a.keys_out = nullptr; a.vals_out = valsB;
            a.g = g; a.shift = 56; a.mask = 255u; a.next_shift = -1; a.next_mask = 0; a.digit_base = ws.base();
            a.next_hist = nullptr; a.status = ws.status; a.ticket = ws.tickets(); a.epoch = ws.epoch; a.dstat = ws.dstat;
            a.home_mode = 0; a.incl_mask = SA_INCL_MASK;
            a.keys_out32 = reinterpret_cast<u32*>(keysB); a.narrow_shift = begin_bit;
            if ((rc = ws.timer.start(stream, 1))) return rc;
            hipLaunchKernelGGL((radix_onesweep_kernel<512, 0, true>), dim3(g.tiles), dim3(512), 0, stream, a);
            if ((rc = ws.timer.stop(stream, (u64)n * 16u))) return rc;
            ws.pass_records += n; ws.pass_bytes += (u64)n * 16u; ws.passes += 1;
        }
    }
    const u32 seg_tile_n = 512u * SEG_ITEMS;
    const u32 flat_max = n / seg_tile_n + 1 + RADIX;   // >= sum over buckets of ceil(size / tile)
    const u32 split_tile_n = 512u * (u32)nw.split_items;   // the split pass has its own tile size: the plan is made for it first
    const u32 split_flat_max = n / split_tile_n + 1 + RADIX;
    hipLaunchKernelGGL(seg_plan_kernel, dim3(1), dim3(256), 0, stream, ws.hist(0), try_split ? split_tile_n : seg_tile_n, nw.plan);
    // narrow keys now in keysB (u32), values in valsB.
    // Three-pass plan (radix_split.hpp): the records of a bucket are grouped by their next rb key bits, then every group is
    // ordered completely in LDS -- when the largest group fits (near-random text; the device's own count decides)
    nw.split_used = false;
    nw.split_flags_done = false;
    nw.split_max_seen = 0;
    if (try_split) {
        const int hb = (lo_bits - LOCAL_BIN_BITS < SPLIT_BITS) ? lo_bits - LOCAL_BIN_BITS : SPLIT_BITS;   // key bits the histogram looks at
        SA_HIP_CHECK(hipMemsetAsync(nw.split_hist, 0, NarrowWorkspace::split_table_bytes(), stream));
        SA_HIP_CHECK(hipMemsetAsync(nw.split_levels_dev(), 0, 16 * sizeof(u32), stream));
        const u32 tpb = 4;
        if (nw.split_items == 32)
            hipLaunchKernelGGL((split_hist_kernel<512, 32>), dim3(div_up(split_flat_max, tpb)), dim3(512), 0, stream,
                               reinterpret_cast<const u32*>(keysB), nw.plan, lo_bits - hb, (1u << hb) - 1u, nw.split_hist, tpb);
        else if (nw.split_items == 28)
            hipLaunchKernelGGL((split_hist_kernel<512, 28>), dim3(div_up(split_flat_max, tpb)), dim3(512), 0, stream,
                               reinterpret_cast<const u32*>(keysB), nw.plan, lo_bits - hb, (1u << hb) - 1u, nw.split_hist, tpb);
        else
            hipLaunchKernelGGL((split_hist_kernel<512, 24>), dim3(div_up(split_flat_max, tpb)), dim3(512), 0, stream,
                               reinterpret_cast<const u32*>(keysB), nw.plan, lo_bits - hb, (1u << hb) - 1u, nw.split_hist, tpb);
        hipLaunchKernelGGL(split_levels_kernel, dim3(RADIX), dim3(SPLIT_NB), 0, stream, (const u32*)nw.split_hist, hb, nw.split_levels_dev());
        SA_HIP_CHECK(hipMemcpyAsync(nw.host_word, nw.split_levels_dev(), 16 * sizeof(u32), hipMemcpyDeviceToHost, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        int rb = -1;
        bool big = nw.local_big;   // sub-buckets of up to LOCAL_CAP_BIG records: one workgroup per CU in the local pass
        if (!big)
            for (int k = 1; k <= hb; ++k)
                if (nw.host_word[k] <= nw.split_cap) { rb = k; break; }
        if (rb < 0 && nw.split_cap == LOCAL_CAP) {
            for (int k = 1; k <= hb; ++k)
                if (nw.host_word[k] <= LOCAL_CAP_BIG && lo_bits - k >= 12) { rb = k; big = true; break; }   // (its 4096 bins: 12 key bits below the sub-bucket's)
            if (rb < 0) big = false;
        }
        nw.split_max_seen = nw.host_word[rb > 0 ? rb : hb];
        if (rb > 0) {
            const int rest_bits = lo_bits - rb;   // >= LOCAL_BIN_BITS
            const int dshift = lo_bits - rb;
            const u32 dmask = (1u << rb) - 1u;
            hipLaunchKernelGGL(split_scan_kernel, dim3(RADIX), dim3(SPLIT_NB), 0, stream, (const u32*)nw.split_hist, nw.plan, hb, rb,
                               nw.split_base, nw.split_sub);
            if (!nw.split_atomic && (rc = nw.ensure_split_status(split_flat_max, stream))) return rc;
            if (nw.split_atomic) SA_HIP_CHECK(hipMemsetAsync(nw.split_cursor, 0, NarrowWorkspace::split_table_bytes(), stream));
            if (++ws.epoch >= (1u << 30)) {
                SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
                ws.epoch = 1;
            }
            if (!nw.split_atomic && ws.epoch <= nw.split_epoch)   // the sort's epoch has wrapped since the last split pass
                SA_HIP_CHECK(hipMemsetAsync(nw.split_status, 0, (size_t)nw.split_tiles * SPLIT_NB * sizeof(u64), stream));
            nw.split_epoch = ws.epoch;
            SplitPassArgs a;
            a.keys_in = reinterpret_cast<const u32*>(keysB); a.vals_in = valsB;
            a.keys_out = reinterpret_cast<u32*>(keysA); a.vals_out = valsA;
            a.plan = nw.plan; a.shift = dshift; a.mask = dmask; a.digit_base = nw.split_base; a.status = nw.split_status;
            a.ticket = nw.tickets; a.epoch = ws.epoch; a.dstat = ws.dstat; a.incl_mask = SA_INCL_MASK; a.cursor = nw.split_cursor;
            a.cur_bs = nw.split_cursor_t ? 1u : (u32)SPLIT_NB; a.cur_ds = nw.split_cursor_t ? (u32)RADIX : 1u;
            if ((rc = ws.timer.start(stream, 2))) return rc;
            if (nw.split_atomic) {
                if (nw.split_items == 32) hipLaunchKernelGGL((seg_split_kernel<512, 32, true>), dim3(split_flat_max), dim3(512), 0, stream, a);
                else if (nw.split_items == 28) hipLaunchKernelGGL((seg_split_kernel<512, 28, true>), dim3(split_flat_max), dim3(512), 0, stream, a);
                else hipLaunchKernelGGL((seg_split_kernel<512, 24, true>), dim3(split_flat_max), dim3(512), 0, stream, a);
            } else {
                if (nw.split_items == 32) hipLaunchKernelGGL((seg_split_kernel<512, 32, false>), dim3(split_flat_max), dim3(512), 0, stream, a);
                else if (nw.split_items == 28) hipLaunchKernelGGL((seg_split_kernel<512, 28, false>), dim3(split_flat_max), dim3(512), 0, stream, a);
                else hipLaunchKernelGGL((seg_split_kernel<512, 24, false>), dim3(split_flat_max), dim3(512), 0, stream, a);
            }
            if ((rc = ws.timer.stop(stream, (u64)n * 16u))) return rc;
            ws.pass_records += n; ws.pass_bytes += (u64)n * 16u; ws.passes += 1;
            LocalArgs l;
            l.keys_in = reinterpret_cast<const u32*>(keysA); l.vals_in = valsA;
            l.keys_out = reinterpret_cast<u32*>(keysB); l.vals_out = valsB; l.vals_out64 = vals_res64;
            l.sub = nw.split_sub;
            const int bb = (big || (rest_bits >= 12 && nw.local_bin_bits == 12)) ? 12 : 11;
            l.bin_shift = rest_bits - bb;
            l.dstat = ws.dstat;
            const u64 local_bytes = (u64)n * 16u + (vals_res64 ? (u64)n * 8u : 0u);
            // the build's first flags pass inside the local pass, when the caller asks for it and provides the buffers
            nw.split_flags_done = false;
            l.dir = DirArgs{}; l.counts = nullptr; l.lite = LiteArgs{}; l.lo_shift = begin_bit; l.rb = rb; l.n = n;
            bool with_flags = false;
            if (nw.split_flags && flags_req && flags_req->prepare) {
                const int fr = flags_req->prepare(flags_req->ctx, (u32)RADIX << rb, &l);
                if (fr < 0) return fr;
                // (the slice of the directory that belongs to a sub-bucket is read off the local pass's bin starts: the bins must be
                //  at least as fine as the directory, and the directory's bits must lie inside the narrow key)
                with_flags = (fr == 0) && l.dir.dir && l.dir.dbits >= 8 + rb && l.dir.dbits - 8 - rb <= bb && l.dir.dbits - 8 <= lo_bits;
            }
            if ((rc = ws.timer.start(stream, 3))) return rc;
            const dim3 lgrid((u32)RADIX << rb), lblock(big ? LOCAL_BLOCK_BIG : LOCAL_BLOCK);
            if (big) {
                if (with_flags) hipLaunchKernelGGL((local_finish_kernel<12, true, LOCAL_BLOCK_BIG>), lgrid, lblock, 0, stream, l);
                else hipLaunchKernelGGL((local_finish_kernel<12, false, LOCAL_BLOCK_BIG>), lgrid, lblock, 0, stream, l);
            } else if (with_flags) {
                if (bb == 12) hipLaunchKernelGGL((local_finish_kernel<12, true>), lgrid, lblock, 0, stream, l);
                else hipLaunchKernelGGL((local_finish_kernel<11, true>), lgrid, lblock, 0, stream, l);
            } else {
                if (bb == 12) hipLaunchKernelGGL((local_finish_kernel<12, false>), lgrid, lblock, 0, stream, l);
                else hipLaunchKernelGGL((local_finish_kernel<11, false>), lgrid, lblock, 0, stream, l);
            }
            nw.split_flags_done = with_flags;
            nw.split_big = big;
            if ((rc = ws.timer.stop(stream, local_bytes))) return rc;
            ws.pass_records += n; ws.pass_bytes += local_bytes; ws.passes += 1;
            SA_HIP_CHECK(hipGetLastError());
            if (diag_env("SA_HIP_SPLIT_DEBUG")) {   // diagnostic: both passes checked on the host
                const u32 nsub = (u32)RADIX << rb;
                std::vector<u32> sub(nsub + 1), k1(n), k2(n), v2(n), bst(RADIX + 1);
                SA_HIP_CHECK(hipStreamSynchronize(stream));
                SA_HIP_CHECK(hipMemcpy(sub.data(), nw.split_sub, (size_t)(nsub + 1) * 4, hipMemcpyDeviceToHost));
                SA_HIP_CHECK(hipMemcpy(k1.data(), keysA, (size_t)n * 4, hipMemcpyDeviceToHost));
                SA_HIP_CHECK(hipMemcpy(k2.data(), keysB, (size_t)n * 4, hipMemcpyDeviceToHost));
                SA_HIP_CHECK(hipMemcpy(v2.data(), valsB, (size_t)n * 4, hipMemcpyDeviceToHost));
                SA_HIP_CHECK(hipMemcpy(bst.data(), nw.plan->bstart, (size_t)(RADIX + 1) * 4, hipMemcpyDeviceToHost));
                u64 bad_sub = 0, bad_split = 0, bad_sorted = 0, bad_mono = 0;
                u32 first_split = ~0u, first_sorted = ~0u;
                for (u32 i = 0; i < nsub; ++i) {
                    if (sub[i] > sub[i + 1] || sub[i + 1] > n) { ++bad_mono; continue; }
                    if ((i & (((u32)1 << rb) - 1)) == 0 && sub[i] != bst[i >> rb]) ++bad_sub;
                    for (u32 j = sub[i]; j < sub[i + 1]; ++j) {
                        if (((k1[j] >> dshift) & dmask) != (i & dmask)) { ++bad_split; if (first_split == ~0u) first_split = j; }
                        if (j > sub[i] && (k2[j - 1] > k2[j] || (k2[j - 1] == k2[j] && v2[j - 1] >= v2[j]))) { ++bad_sorted; if (first_sorted == ~0u) first_sorted = j; }
                    }
                }
                fprintf(stderr, "[sa_hip] split debug: n=%u rb=%d hb=%d nsub=%u sub[0]=%u sub[last]=%u | table: %llu not monotone, %llu off the bucket starts | split pass: %llu records in the wrong group (first %u) | local pass: %llu out of order (first %u)\n",
                        n, rb, hb, nsub, sub[0], sub[nsub], (unsigned long long)bad_mono, (unsigned long long)bad_sub, (unsigned long long)bad_split, first_split,
                        (unsigned long long)bad_sorted, first_sorted);
            }
            nw.split_used = true;
            nw.split_rb = rb;
            *keys_res = reinterpret_cast<u64*>(keysB);
            *vals_res = valsB;
            return 0;
        }
    }
    if (optimistic) {   // declined: the LSD passes keep ties in the order they find them, so the top-digit pass once more, stable
        SA_HIP_CHECK(hipMemsetAsync(ws.tickets(), 0, NCHUNK * sizeof(u32), stream));
        if (++ws.epoch >= (1u << 30)) {
            SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
            ws.epoch = 1;
        }
        if ((rc = launch_text_top(false))) return rc;
    }
    if (try_split && split_tile_n != seg_tile_n)   // declined: the plan again, for the tiles of the LSD passes
        hipLaunchKernelGGL(seg_plan_kernel, dim3(1), dim3(256), 0, stream, ws.hist(0), seg_tile_n, nw.plan);
    // histogram of the first narrow digit per bucket (the LSD passes' histograms are zeroed here: the three-pass plan has no use for them)
    SA_HIP_CHECK(hipMemsetAsync(nw.hist, 0, NarrowWorkspace::hist_bytes(), stream));
    {
        const u32 tpb = 4;
        const u32 mask0 = (1u << ((np == 1) ? lo_bits : RADIX_BITS)) - 1u;
        hipLaunchKernelGGL((seg_hist_kernel<512, SEG_ITEMS>), dim3(div_up(flat_max, tpb)), dim3(512), 0, stream,
                           reinterpret_cast<const u32*>(keysB), nw.plan, 0, mask0, nw.hist, tpb);
    }
    u32* kin = reinterpret_cast<u32*>(keysB); u32* vin = valsB;
    u32* kout = reinterpret_cast<u32*>(keysA); u32* vout = valsA;
    for (int p = 0; p < np; ++p) {
        if (++ws.epoch >= (1u << 30)) {
            SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
            ws.epoch = 1;
        }
        const bool last = (p == np - 1);
        const int bits_p = last ? (lo_bits - RADIX_BITS * (np - 1)) : RADIX_BITS;
        hipLaunchKernelGGL(seg_scan_kernel, dim3(RADIX), dim3(256), 0, stream, nw.hist + (size_t)p * RADIX * RADIX, nw.plan, nw.base);
        SegPassArgs a;
        a.keys_in = kin; a.vals_in = vin; a.keys_out = kout; a.keys_out64 = keep_narrow ? nullptr : reinterpret_cast<u64*>(kout);
        a.vals_out = vout;
        a.vals_out64 = last ? vals_res64 : nullptr;
        a.plan = nw.plan;
        a.shift = RADIX_BITS * p;
        a.mask = (1u << bits_p) - 1u;
        a.next_shift = last ? -1 : RADIX_BITS * (p + 1);
        const int bits_n = (p + 1 == np - 1) ? (lo_bits - RADIX_BITS * (np - 1)) : RADIX_BITS;
        a.next_mask = last ? 0u : ((1u << bits_n) - 1u);
        a.digit_base = nw.base;
        a.next_hist = last ? nullptr : nw.hist + (size_t)(p + 1) * RADIX * RADIX;
        a.status = ws.status; a.ticket = nw.tickets + p * NCHUNK; a.epoch = ws.epoch; a.dstat = ws.dstat;
        a.lo_shift = begin_bit; a.incl_mask = SA_INCL_MASK;   // every 4th tile: 1 / 3 / 7 / 15 measured 3.84 / 3.74 / 3.80 / 3.98 ms per pass
        if ((rc = ws.timer.start(stream, last ? 3 : 2))) return rc;
        if (last) hipLaunchKernelGGL((seg_onesweep_kernel<512, SEG_ITEMS, true>), dim3(flat_max), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((seg_onesweep_kernel<512, SEG_ITEMS, false>), dim3(flat_max), dim3(512), 0, stream, a);
        const u64 pass_bytes = (u64)n * ((last && !keep_narrow) ? 20u : 16u) + ((last && vals_res64) ? (u64)n * 8u : 0u);
        if ((rc = ws.timer.stop(stream, pass_bytes))) return rc;
        ws.pass_records += n; ws.pass_bytes += pass_bytes; ws.passes += 1;
        u32* tk = kin; kin = kout; kout = tk;
        u32* tv = vin; vin = vout; vout = tv;
    }
    SA_HIP_CHECK(hipGetLastError());
    *keys_res = reinterpret_cast<u64*>(kin);
    *vals_res = vin;
    return 0;
}

}  // namespace sa
