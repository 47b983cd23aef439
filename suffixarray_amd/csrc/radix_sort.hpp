// radix_sort.hpp -- hand-written LSD radix sort of (u64 key, u32 value) records for gfx950.
//
// One-sweep structure: ONE histogram pre-pass over all digits, then per 8-bit digit ONE kernel
// that reads every record once and writes it once.  Inside a pass each workgroup
//   1. takes the next tile id from an atomic ticket (tiles start in ticket order, so every
//      predecessor tile is already resident -> look-back cannot deadlock),
//   2. loads its tile wave-striped (64 lanes x 8 B = 512 B per load instruction),
//   3. ranks its keys per wave with __ballot match masks (64-wide; 8 ballots per key) and
//      per-wave digit counters in LDS -- no LDS atomics, stable by construction,
//   4. publishes its 256 digit counts as 8-byte {epoch,flag,count} granules (one relaxed
//      agent-scope store each; the data IS the flag, cdna_hip_programming.md G16/R2) and
//      resolves its exclusive prefix by decoupled look-back over predecessor tiles
//      (relaxed agent-scope loads; every spin is bounded and sets DeviceStatus.error),
//   5. reorders keys (then values) through LDS so that each digit's run leaves the CU as
//      contiguous global stores.
// Algorithmic bytes per pass over M records: 2*M*(8+4); the pre-pass adds 8*M.
//
// This replaces, by function only, the bucket placement / induced-sorting scans of the
// reference's libsais (libsais.c:1542-1614, 2110-2141, 2942-2975): same output order, no
// shared code or structure.
#pragma once
#include "common.hpp"

namespace sa {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_BLOCK = 256;   // 4 waves
constexpr int SORT_ITEMS = 16;    // records per thread
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;  // 4096 records = 48 KB of (key,value)
constexpr u32 SPIN_LIMIT = 1u << 22;

// tile status granule: [63:34] epoch | [33:32] flag | [31:0] count
constexpr u64 FLAG_AGG = 1, FLAG_INCL = 2;
__device__ __forceinline__ u64 pack_status(u32 epoch, u64 flag, u32 v) {
    return ((u64)epoch << 34) | (flag << 32) | (u64)v;
}

struct SortPassArgs {
    const u64* keys_in;
    const u32* vals_in;   // nullptr: value = record position (iota), first pass of a build
    u64* keys_out;
    u32* vals_out;
    u32 n;
    int shift;
    u32 mask;
    const u32* digit_base;  // [RADIX] exclusive global offsets of this pass
    u64* status;            // [tiles][RADIX]
    u32* ticket;            // tile ticket counter of this pass (zeroed by the host)
    u32 epoch;
    DeviceStatus* dstat;
};

// ---- histogram pre-pass: all digits of [begin_bit, begin_bit + 8*npasses) at once -------------
__global__ __launch_bounds__(256) void radix_hist_kernel(const u64* __restrict__ keys, u32 n, int begin_bit,
                                                         int npasses, u32 last_mask, u32* __restrict__ ghist) {
    __shared__ u32 s_h[8 * RADIX];
    for (int i = threadIdx.x; i < npasses * RADIX; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 k = keys[i];
        for (int p = 0; p < npasses; ++p) {
            u32 d = (u32)(k >> (begin_bit + RADIX_BITS * p)) & (RADIX - 1);
            if (p == npasses - 1) d &= last_mask;
            atomicAdd(&s_h[p * RADIX + d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npasses * RADIX; i += blockDim.x) {
        const u32 v = s_h[i];
        if (v) atomicAdd(&ghist[i], v);
    }
}

// exclusive scan of each pass's 256 bins; grid = npasses, block = 256
__global__ __launch_bounds__(256) void radix_scan_hist_kernel(const u32* __restrict__ ghist, u32* __restrict__ gbase) {
    __shared__ u32 s_w[4];
    const int d = threadIdx.x, lane = d & 63, w = d >> 6;
    const u32 c = ghist[blockIdx.x * RADIX + d];
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    u32 woff = 0;
    for (int i = 0; i < w; ++i) woff += s_w[i];
    gbase[blockIdx.x * RADIX + d] = woff + incl - c;
}

// ---- the pass -----------------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_BLOCK) void radix_onesweep_kernel(SortPassArgs a) {
    constexpr int WAVES = SORT_BLOCK / WAVE;
    __shared__ __attribute__((aligned(16))) u64 s_keys[SORT_TILE];  // reused as u32 values afterwards
    __shared__ u32 s_whist[WAVES * RADIX];
    __shared__ u32 s_dstart[RADIX];
    __shared__ u32 s_gdelta[RADIX];
    __shared__ u32 s_wsum[WAVES];
    __shared__ u32 s_tile;
    __shared__ u32 s_abort;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_tile = atomicAdd(a.ticket, 1u);
        // a failed spin anywhere poisons the sort: later tiles drain instead of spinning again
        s_abort = __hip_atomic_load(&a.dstat->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = tid; i < WAVES * RADIX; i += SORT_BLOCK) s_whist[i] = 0;
    __syncthreads();
    const u32 tile = s_tile;
    const u64 tile_base = (u64)tile * SORT_TILE;
    if (tile_base >= a.n || s_abort != 0) return;  // block-uniform

    // 1. load (wave-striped): wave w owns records [w*64*ITEMS, (w+1)*64*ITEMS) of the tile
    u64 key[SORT_ITEMS];
    u32 val[SORT_ITEMS];
    const u64 wbase = tile_base + (u64)wave * (WAVE * SORT_ITEMS) + lane;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const u64 p = wbase + (u64)j * WAVE;
        key[j] = (p < a.n) ? a.keys_in[p] : ~0ull;
    }
    if (a.vals_in) {
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const u64 p = wbase + (u64)j * WAVE;
            val[j] = (p < a.n) ? a.vals_in[p] : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) val[j] = (u32)(wbase + (u64)j * WAVE);
    }

    // 2. per-wave stable ranking with ballot match masks
    u32 rank[SORT_ITEMS];
    volatile u32* wh = s_whist + wave * RADIX;
    const u64 lt = lanemask_lt();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const bool valid = (wbase + (u64)j * WAVE) < a.n;
        const u32 d = (u32)(key[j] >> a.shift) & a.mask;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const u32 below = (u32)__popcll(peers & lt);
        const u32 prior = valid ? wh[d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) wh[d] = prior + (u32)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[j] = prior + below;
    }
    __syncthreads();

    // 3. tile digit counts -> publish aggregate -> exclusive scan over digits -> look-back
    u32 count = 0, excl = 0;
    {
        // SORT_BLOCK == RADIX: thread d owns digit d
        u32 c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const u32 t = s_whist[w * RADIX + tid];
            s_whist[w * RADIX + tid] = c;  // exclusive over waves
            c += t;
        }
        count = c;
        __hip_atomic_store(&a.status[(u64)tile * RADIX + tid],
                           pack_status(a.epoch, tile == 0 ? FLAG_INCL : FLAG_AGG, count),
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
    {
        u32 woff = 0;
        for (int i = 0; i < wave; ++i) woff += s_wsum[i];
        excl += woff;
        s_dstart[tid] = excl;
        u32 prefix = 0;
        if (tile > 0) {
            int64_t t = (int64_t)tile - 1;
            bool dead = false;
            while (true) {
                u64 w;
                u32 spins = 0;
                while (true) {
                    w = __hip_atomic_load(&a.status[(u64)t * RADIX + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((u32)(w >> 34) == a.epoch && ((w >> 32) & 3u) != 0) break;
                    ++spins;
                    if ((spins & 1023u) == 0) {
                        if (spins >= SPIN_LIMIT ||
                            __hip_atomic_load(&a.dstat->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            __hip_atomic_store(&a.dstat->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            dead = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (dead) break;
                prefix += (u32)w;
                if (((w >> 32) & 3u) == FLAG_INCL || t == 0) break;
                --t;
            }
            __hip_atomic_store(&a.status[(u64)tile * RADIX + tid], pack_status(a.epoch, FLAG_INCL, prefix + count),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gdelta[tid] = a.digit_base[tid] + prefix - excl;
    }
    __syncthreads();

    // 4. keys -> LDS at their tile-local sorted position -> coalesced global stores per digit run
    u32 pos[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const bool valid = (wbase + (u64)j * WAVE) < a.n;
        const u32 d = (u32)(key[j] >> a.shift) & a.mask;
        pos[j] = s_dstart[d] + s_whist[wave * RADIX + d] + rank[j];
        if (valid) s_keys[pos[j]] = key[j];
    }
    __syncthreads();
    const u32 tile_n = (u32)((a.n - tile_base) < (u64)SORT_TILE ? (a.n - tile_base) : (u64)SORT_TILE);
    u32 gidx[SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const u32 p = k * SORT_BLOCK + tid;
        if (p < tile_n) {
            const u64 kk = s_keys[p];
            const u32 d = (u32)(kk >> a.shift) & a.mask;
            gidx[k] = s_gdelta[d] + p;
            a.keys_out[gidx[k]] = kk;
        }
    }
    __syncthreads();
    u32* s_vals = reinterpret_cast<u32*>(s_keys);
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const bool valid = (wbase + (u64)j * WAVE) < a.n;
        if (valid) s_vals[pos[j]] = val[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const u32 p = k * SORT_BLOCK + tid;
        if (p < tile_n) a.vals_out[gidx[k]] = s_vals[p];
    }
}

static_assert(SORT_BLOCK == RADIX, "one thread per digit in the scan / look-back phase");

// ---- host driver ----------------------------------------------------------------------------------

// HIP-event stopwatch for one class of launches (accumulated per build / per batch).
struct EventTimer {
    static constexpr int CAP = 128;
    hipEvent_t ev[2 * CAP];
    int used = 0;
    bool ready = false;
    double total_ms = 0.0;
    u64 launches = 0;

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
    void reset() { used = 0; total_ms = 0.0; launches = 0; }
    int flush() {
        for (int i = 0; i < used; ++i) {
            SA_HIP_CHECK(hipEventSynchronize(ev[2 * i + 1]));
            float ms = 0.f;
            SA_HIP_CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            total_ms += ms;
        }
        used = 0;
        return 0;
    }
    int start(hipStream_t s) {
        if (used == CAP) { int rc = flush(); if (rc) return rc; }
        SA_HIP_CHECK(hipEventRecord(ev[2 * used], s));
        return 0;
    }
    int stop(hipStream_t s) {
        SA_HIP_CHECK(hipEventRecord(ev[2 * used + 1], s));
        ++used;
        ++launches;
        return 0;
    }
};

struct RadixWorkspace {
    u64* status = nullptr;     // [max_tiles][RADIX]
    u32* small = nullptr;      // tickets[16] | ghist[8*RADIX] | gbase[8*RADIX]
    DeviceStatus* dstat = nullptr;
    u32 max_tiles = 0;
    u32 epoch = 0;
    EventTimer timer;          // onesweep launches only
    u64 pass_records = 0;      // sum over passes of records moved
    u64 passes = 0;

    u32* tickets() const { return small; }
    u32* ghist() const { return small + 16; }
    u32* gbase() const { return small + 16 + 8 * RADIX; }
    static size_t small_bytes() { return (16 + 16 * RADIX) * sizeof(u32); }

    int init(u64 n_max) {
        max_tiles = div_up(n_max ? n_max : 1, SORT_TILE);
        SA_HIP_CHECK(hipMalloc(&status, (size_t)max_tiles * RADIX * sizeof(u64)));
        SA_HIP_CHECK(hipMemset(status, 0, (size_t)max_tiles * RADIX * sizeof(u64)));
        SA_HIP_CHECK(hipMalloc(&small, small_bytes()));
        SA_HIP_CHECK(hipMalloc(&dstat, sizeof(DeviceStatus)));
        SA_HIP_CHECK(hipMemset(dstat, 0, sizeof(DeviceStatus)));
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
    void reset_stats() { timer.reset(); pass_records = 0; passes = 0; }
};

// Sort n records by key bits [begin_bit, end_bit), stable.  Buffers ping-pong A -> B -> A ...;
// on return *keys_res / *vals_res point at the buffers holding the result.  vals iota: the
// first pass generates value = position instead of reading valsA.
inline int radix_sort_pairs(RadixWorkspace& ws, hipStream_t stream, u64* keysA, u32* valsA, u64* keysB, u32* valsB,
                            u32 n, int begin_bit, int end_bit, bool iota_vals, u64** keys_res, u32** vals_res) {
    *keys_res = keysA;
    *vals_res = valsA;
    if (n == 0 || end_bit <= begin_bit) {
        if (iota_vals && n) return fail(SA_HIP_EINVAL, "radix_sort_pairs: iota with zero passes");
        return 0;
    }
    const int npasses = (end_bit - begin_bit + RADIX_BITS - 1) / RADIX_BITS;
    if (npasses > 8) return fail(SA_HIP_EINVAL, "radix_sort_pairs: more than 8 passes");
    const int last_bits = end_bit - begin_bit - RADIX_BITS * (npasses - 1);
    const u32 last_mask = (1u << last_bits) - 1u;
    const u32 tiles = div_up(n, SORT_TILE);
    if (tiles > ws.max_tiles) return fail(SA_HIP_EINVAL, "radix_sort_pairs: workspace too small");

    SA_HIP_CHECK(hipMemsetAsync(ws.small, 0, RadixWorkspace::small_bytes(), stream));
    u32 hgrid = div_up(n, 256 * 16);
    if (hgrid > 2048) hgrid = 2048;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(hgrid), dim3(256), 0, stream, keysA, n, begin_bit, npasses, last_mask,
                       ws.ghist());
    hipLaunchKernelGGL(radix_scan_hist_kernel, dim3(npasses), dim3(256), 0, stream, ws.ghist(), ws.gbase());

    u64* kin = keysA; u32* vin = valsA; u64* kout = keysB; u32* vout = valsB;
    for (int p = 0; p < npasses; ++p) {
        if (++ws.epoch >= (1u << 30)) {  // epoch wrap: re-zero the granules once per 2^30 passes
            SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
            ws.epoch = 1;
        }
        SortPassArgs a;
        a.keys_in = kin;
        a.vals_in = (p == 0 && iota_vals) ? nullptr : vin;
        a.keys_out = kout;
        a.vals_out = vout;
        a.n = n;
        a.shift = begin_bit + RADIX_BITS * p;
        a.mask = (p == npasses - 1) ? last_mask : (u32)(RADIX - 1);
        a.digit_base = ws.gbase() + p * RADIX;
        a.status = ws.status;
        a.ticket = ws.tickets() + p;
        a.epoch = ws.epoch;
        a.dstat = ws.dstat;
        int rc = ws.timer.start(stream);
        if (rc) return rc;
        hipLaunchKernelGGL(radix_onesweep_kernel, dim3(tiles), dim3(SORT_BLOCK), 0, stream, a);
        rc = ws.timer.stop(stream);
        if (rc) return rc;
        ws.pass_records += n;
        ws.passes += 1;
        u64* tk = kin; kin = kout; kout = tk;
        u32* tv = vin; vin = vout; vout = tv;
    }
    SA_HIP_CHECK(hipGetLastError());
    *keys_res = kin;
    *vals_res = vin;
    return 0;
}

}  // namespace sa
