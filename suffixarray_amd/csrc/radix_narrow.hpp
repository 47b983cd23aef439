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
//   top digit       u64 key          u32 low key, u32 value (iota)     8 + 8
//   histogram       u32 low key      -                                 4        (digit 0 per bucket)
//   narrow passes   u32 key, u32 val u32 key, u32 val                  8 + 8
//   last pass       u32 key, u32 val u64 key (rebuilt), u32 val        8 + 12
//
// 40-bit keys (the build's choice for near-random text: 8 characters of 5 bits): 16 + 4 + 3*16 + 20 = 88
// bytes per record instead of 20 + 4*24 = 116.
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

namespace sa {

constexpr int NARROW_MAX_PASSES = 4;

struct SegPlan {              // device resident, written by seg_plan_kernel
    u32 bstart[RADIX + 1];    // first record of bucket b; [RADIX] = n
    u32 tprefix[RADIX + 1];   // flat index of the first tile of bucket b; [RADIX] = number of tiles
    u32 cfirst[NCHUNK + 1];   // ticket ranges: part c = flat tiles [cfirst[c], cfirst[c+1]), whole buckets each
};

// bucket sizes from the per-chunk histogram of the top digit -> SegPlan (one workgroup of 256)
__global__ __launch_bounds__(256) void seg_plan_kernel(const u32* __restrict__ hist_top, u32 tile_shift, SegPlan* __restrict__ plan) {
    __shared__ u32 s_w[2][4];
    __shared__ u32 s_tp[RADIX + 1];
    const int d = threadIdx.x, lane = d & 63, w = d >> 6;
    u32 size = 0;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) size += hist_top[c * RADIX + d];
    const u32 tiles = (size + (1u << tile_shift) - 1) >> tile_shift;
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

// histogram of digit [shift, shift + 8) of the narrow keys, per bucket: hist[b][d]
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void seg_hist_kernel(const u32* __restrict__ keys, const SegPlan* __restrict__ plan, int shift,
                                                         u32 mask, u32* __restrict__ hist, u32 tiles_per_block) {
    constexpr u32 TILE = BLOCK * SORT_ITEMS;
    __shared__ u32 s_h[RADIX];
    __shared__ u32 s_t[RADIX + 1];
    for (int i = threadIdx.x; i <= RADIX; i += BLOCK) s_t[i] = plan->tprefix[i];
    for (int i = threadIdx.x; i < RADIX; i += BLOCK) s_h[i] = 0;
    __syncthreads();
    const u32 F = s_t[RADIX];
    const u32 f_lo = blockIdx.x * tiles_per_block;
    const u32 f_hi = (f_lo + tiles_per_block < F) ? f_lo + tiles_per_block : F;
    if (f_lo >= f_hi) return;
    u32 cur = seg_bucket_of(s_t, f_lo);
    for (u32 f = f_lo; f < f_hi; ++f) {
        const u32 b = seg_bucket_of(s_t, f);
        if (b != cur) { hist_flush(s_h, hist, cur); cur = b; }
        const u32 start = plan->bstart[b] + ((f - s_t[b]) * TILE);
        const u32 end = plan->bstart[b + 1];
        const u32 len = (end - start) < TILE ? (end - start) : TILE;
        for (u32 l = threadIdx.x; l < len; l += BLOCK) atomicAdd(&s_h[(keys[start + l] >> shift) & mask], 1u);
    }
    hist_flush(s_h, hist, cur);
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
    u64* keys_out64;       // LAST: (bucket << 56) | (narrow key << lo_shift)
    u32* vals_out;
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

template <bool FULL, int BLOCK, bool LAST>
__device__ __forceinline__ void seg_tile(const SegPassArgs& a, const u32 flat, const u32 first_flat, const u32 bucket,
                                         const u32 start, const u32 tile_n, u32* s_keys, u32* s_whist, u32* s_gdelta,
                                         u32* s_wsum) {
    constexpr int WAVES = BLOCK / WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 woff = (u32)wave * (WAVE * SORT_ITEMS) + lane;

    // 1. load (wave-striped)
    u32 key[SORT_ITEMS];
    const u32* kin = a.keys_in + start;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        key[j] = (FULL || p < tile_n) ? kin[p] : ~0u;
    }
    // 2. rank
    u32 rd[SORT_ITEMS];
    u32* wh = s_whist + wave * RADIX;
    wave_rank<FULL>(key, a.shift, a.mask, woff, tile_n, wh, rd);
    u32 val[SORT_ITEMS];
    const u32* vin = a.vals_in + start;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        val[j] = (FULL || p < tile_n) ? vin[p] : 0u;
    }
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
    u32 pos[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
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
        s_gdelta[tid] = a.digit_base[bucket * RADIX + tid] + prefix - excl;
    }
    __syncthreads();

    // 6. coalesced stores per digit run
    u32 gidx[SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const u32 p = k * BLOCK + tid;
        if (FULL || p < tile_n) {
            const u32 kk = s_keys[p];
            gidx[k] = s_gdelta[(kk >> a.shift) & a.mask] + p;
            if (LAST) a.keys_out64[gidx[k]] = ((u64)bucket << 56) | ((u64)kk << a.lo_shift);
            else {
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
    u32* s_vals = s_keys;
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

template <int BLOCK, bool LAST>
__global__ __launch_bounds__(BLOCK, 4) void seg_onesweep_kernel(SegPassArgs a) {
    constexpr int WAVES = BLOCK / WAVE;
    constexpr u32 TILE = BLOCK * SORT_ITEMS;
    __shared__ __attribute__((aligned(16))) u32 s_keys[TILE];   // reused for the values
    __shared__ u32 s_whist[WAVES * RADIX];
    __shared__ u32 s_gdelta[RADIX];
    __shared__ u32 s_wsum[RADIX / WAVE];
    __shared__ u32 s_t[RADIX + 1];
    __shared__ u32 s_c[NCHUNK + 1];
    __shared__ u32 s_flat;

    const int tid = threadIdx.x;
    // one tile per workgroup: the ticket of this XCD's part is requested first, the plan is read meanwhile
    u32 home = 0, t_home = 0;
    if (tid == 0) {
        home = xcc_id();
        t_home = atomicAdd(&a.ticket[home], 1u);
    }
    for (int i = tid; i <= RADIX; i += BLOCK) s_t[i] = a.plan->tprefix[i];
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
    const u32 start = a.plan->bstart[bucket] + (flat - first_flat) * TILE;
    const u32 rest = a.plan->bstart[bucket + 1] - start;
    if (rest >= TILE)
        seg_tile<true, BLOCK, LAST>(a, flat, first_flat, bucket, start, TILE, s_keys, s_whist, s_gdelta, s_wsum);
    else
        seg_tile<false, BLOCK, LAST>(a, flat, first_flat, bucket, start, rest, s_keys, s_whist, s_gdelta, s_wsum);
}

// ---- host driver --------------------------------------------------------------------------------------
struct NarrowWorkspace {
    SegPlan* plan = nullptr;
    u32* hist = nullptr;     // [NARROW_MAX_PASSES][RADIX][RADIX]
    u32* base = nullptr;     // [RADIX][RADIX]
    u32* tickets = nullptr;  // [NARROW_MAX_PASSES][NCHUNK]
    static size_t hist_bytes() { return (size_t)NARROW_MAX_PASSES * RADIX * RADIX * sizeof(u32); }
    int init() {
        SA_HIP_CHECK(hipMalloc(&plan, sizeof(SegPlan)));
        SA_HIP_CHECK(hipMalloc(&hist, hist_bytes()));
        SA_HIP_CHECK(hipMalloc(&base, (size_t)RADIX * RADIX * sizeof(u32)));
        SA_HIP_CHECK(hipMalloc(&tickets, (size_t)NARROW_MAX_PASSES * NCHUNK * sizeof(u32)));
        return 0;
    }
    void destroy() {
        if (plan) (void)hipFree(plan);
        if (hist) (void)hipFree(hist);
        if (base) (void)hipFree(base);
        if (tickets) (void)hipFree(tickets);
        plan = nullptr; hist = nullptr; base = nullptr; tickets = nullptr;
    }
};

// Keys of at most 40 bits and a tile size of 8192; below ~4M records the flat tile count (n / tile + 256)
// would not fit the status array sized for the plain sort, and nothing is to be gained there anyway.
inline bool narrow_sort_applies(const RadixWorkspace& ws, u64 n, int begin_bit) {
    return ws.block == 512 && begin_bit >= 24 && begin_bit < 56 && n >= (1u << 22) &&
           n / ws.tile() + RADIX + 1 <= ws.max_tiles;
}

// Sort n records (keysA[i], i) by key bits [begin_bit, 64), stable.  keysA holds the u64 keys and the
// histogram of their top digit has been accumulated into ws.hist(0) by the producer (radix_prepare() before).
// keysA / keysB and valsA / valsB are the ping-pong buffers of the plain sort (n * 8 and n * 4 bytes);
// narrow keys use the first n * 4 bytes of a key buffer.  Result: *keys_res (u64, full keys), *vals_res.
inline int radix_sort_narrow(RadixWorkspace& ws, NarrowWorkspace& nw, hipStream_t stream, u64* keysA, u32* valsA, u64* keysB,
                             u32* valsB, u32 n, int begin_bit, u64** keys_res, u32** vals_res) {
    int rc;
    const int lo_bits = 56 - begin_bit;                       // 1 .. 32
    const int np = (lo_bits + RADIX_BITS - 1) / RADIX_BITS;   // narrow passes, 1 .. 4
    const SortGeom g = make_geom(n, ws.tile());
    SA_HIP_CHECK(hipMemsetAsync(nw.hist, 0, NarrowWorkspace::hist_bytes(), stream));
    SA_HIP_CHECK(hipMemsetAsync(nw.tickets, 0, (size_t)NARROW_MAX_PASSES * NCHUNK * sizeof(u32), stream));

    // top digit: ordinary one-sweep pass, values generated, keys leave as their low bits
    {
        if (++ws.epoch >= (1u << 30)) {
            SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
            ws.epoch = 1;
        }
        hipLaunchKernelGGL(radix_scan_hist_kernel, dim3(1), dim3(256), 0, stream, ws.hist(0), ws.base());
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
    hipLaunchKernelGGL(seg_plan_kernel, dim3(1), dim3(256), 0, stream, ws.hist(0), g.tile_shift, nw.plan);
    const u32 flat_max = g.tiles + RADIX;   // >= sum over buckets of ceil(size / tile)
    // narrow keys now in keysB (u32), values in valsB; histogram of the first narrow digit per bucket
    {
        const u32 tpb = 4;
        const u32 mask0 = (1u << ((np == 1) ? lo_bits : RADIX_BITS)) - 1u;
        hipLaunchKernelGGL((seg_hist_kernel<512>), dim3(div_up(flat_max, tpb)), dim3(512), 0, stream,
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
        a.keys_in = kin; a.vals_in = vin; a.keys_out = kout; a.keys_out64 = reinterpret_cast<u64*>(kout); a.vals_out = vout;
        a.plan = nw.plan;
        a.shift = RADIX_BITS * p;
        a.mask = (1u << bits_p) - 1u;
        a.next_shift = last ? -1 : RADIX_BITS * (p + 1);
        const int bits_n = (p + 1 == np - 1) ? (lo_bits - RADIX_BITS * (np - 1)) : RADIX_BITS;
        a.next_mask = last ? 0u : ((1u << bits_n) - 1u);
        a.digit_base = nw.base;
        a.next_hist = last ? nullptr : nw.hist + (size_t)(p + 1) * RADIX * RADIX;
        a.status = ws.status; a.ticket = nw.tickets + p * NCHUNK; a.epoch = ws.epoch; a.dstat = ws.dstat;
        a.lo_shift = begin_bit; a.incl_mask = SA_INCL_MASK;
        if ((rc = ws.timer.start(stream, last ? 3 : 2))) return rc;
        if (last) hipLaunchKernelGGL((seg_onesweep_kernel<512, true>), dim3(flat_max), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((seg_onesweep_kernel<512, false>), dim3(flat_max), dim3(512), 0, stream, a);
        if ((rc = ws.timer.stop(stream, (u64)n * (last ? 20u : 16u)))) return rc;
        ws.pass_records += n; ws.pass_bytes += (u64)n * (last ? 20u : 16u); ws.passes += 1;
        u32* tk = kin; kin = kout; kout = tk;
        u32* tv = vin; vin = vout; vout = tv;
    }
    SA_HIP_CHECK(hipGetLastError());
    *keys_res = reinterpret_cast<u64*>(kin);
    *vals_res = vin;
    return 0;
}

}  // namespace sa
