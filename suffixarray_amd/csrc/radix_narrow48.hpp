// radix_narrow48.hpp -- narrow-record sort for keys of 41 .. 56 bits: 10-byte instead of 12-byte records.
//
// Word / name / log-like texts take initial keys of 11-12 characters (55-60 bits, sa_build.hpp: the pilot); until round 3
// those were sorted as (u64 key, u32 suffix) records: 1 + 12 bytes for the text-sourced pass 0 and 24 bytes for each of
// the 7 passes after it -- 35 of the 68 ms of BASELINE config 5.  The narrow-record idea of radix_narrow.hpp carries
// over: sort the TOP digit first (straight from the text) and a record's top digit is given by its bucket; what is
// left of a 56-bit key are 48 bits, kept as a u32 (the upper 32) and a u16 (the lower 16) next to the u32 suffix index:
//
//   pass             reads                     writes                          bytes / record
//   top digit        text (1 byte)             u32 k32, u16 k16, u32 value     1 + 10      text_top_pass_kernel<512, true>
//   histogram        u16 k16                   -                               2           digit 0 per bucket
//   passes 0, 1      k16 (ranked), k32, value  the same                        10 + 10     seg48_onesweep_kernel<.., ON16 = true>
//   passes 2 ..      k32 (ranked), k16, value  the same                        10 + 10     seg48_onesweep_kernel<.., ON16 = false>
//   last pass        k32 (ranked), k16, value  u64 key (rebuilt), u32 value    10 + 12     (+ 8: int64 copy of a 64-bit build)
//
// 55-bit keys (11 characters of 5 bits): 1 + 11 + 2 + 5 * 20 + 22 = 136 bytes per character instead of 1 + 13 + 6 * 24 = 158
// for the same key length on 12-byte records (181 for the 12 characters the old plan took).  Geometry, look-back chains
// (one per bucket), per-bucket digit bases and the next pass's histogram are those of radix_narrow.hpp; the rank is
// computed on whichever array holds the pass's digit, both key arrays are staged in LDS side by side at the record's
// tile-local sorted position (one round for the 6 key bytes), the values follow through the same buffer.  The last pass
// rebuilds full u64 keys -- the flags pass, the bucket directory and the query kernel see what the 12-byte plan left them.
#pragma once
#include "radix_narrow.hpp"

namespace sa {

// records per thread: 14 -- tiles of 7168, k32 and k16 staged side by side in LDS (42 KB of 52 KB), 79 registers: THREE workgroups
// per CU (210 KB of records in flight).  16 (two workgroups, 160 KB in flight) and 20 (two, 200 KB): the sort of the config-5
// column at 4e8 characters 14.16 / 13.80 ms against 13.62 (tools/gpu_lib_ab.py, -DSA_SEG48_ITEMS).
#ifndef SA_SEG48_ITEMS
#define SA_SEG48_ITEMS 14
#endif
constexpr int SEG48_ITEMS = SA_SEG48_ITEMS;
constexpr int SEG48_LAST_ITEMS = SEG48_ITEMS;

struct Seg48Args {
    const u32* k32_in; const u16* k16_in; const u32* vals_in;
    u32* k32_out; u16* k16_out; u32* vals_out;
    u64* keys_out64;       // LAST: (bucket << 56) | (k32 << (begin_bit + 16)) | (k16 << begin_bit)
    int64_t* vals_out64;   // LAST, may be null: the values also as int64 (libsais64 layout, libsais64.c:6248-6259)
    const SegPlan* plan;
    int shift; u32 mask;                 // digit of the array this pass ranks by (ON16: k16, else k32)
    int next_shift; u32 next_mask;       // < 0: last pass
    int next_on16;                       // the next pass's digit lives in k16 (else k32)
    const u32* digit_base; u32* next_hist; u64* status; u32* ticket; u32 epoch; DeviceStatus* dstat;
    int begin_bit; u32 incl_mask;
};

template <bool FULL, int BLOCK, int ITEMS, bool ON16, bool LAST>
__device__ __forceinline__ void seg48_tile(const Seg48Args& a, const u32 flat, const u32 first_flat, const u32 bucket,
                                           const u32 start, const u32 tile_n, u32* s_keys, u16* s_ext, u32* s_whist, u32* s_gdelta,
                                           u32* s_wsum) {
    static_assert(!(LAST && ON16), "the last pass ranks by the upper key bits");
    constexpr int WAVES = BLOCK / WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 woff = (u32)wave * (WAVE * ITEMS) + lane;
    const u32 dbase = (tid < RADIX) ? a.digit_base[bucket * RADIX + tid] : 0u;   // needed after the look-back

    // 1. load the array this pass ranks by (wave-striped)
    u32 key[ITEMS];
    const u32* k32in = a.k32_in + start;
    const u16* k16in = a.k16_in + start;
    const u32* vin = a.vals_in + start;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        if (ON16) key[j] = (FULL || p < tile_n) ? (u32)k16in[p] : 0xFFFFu;
        else key[j] = (FULL || p < tile_n) ? k32in[p] : ~0u;
    }
    // 2. rank
    u32 rd[ITEMS];
    u32* wh = s_whist + wave * RADIX;
    wave_rank<FULL>(key, a.shift, a.mask, woff, tile_n, wh, rd);
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
        __hip_atomic_store(&a.status[(u64)flat * RADIX + tid], pack_status(a.epoch, flat == first_flat ? FLAG_INCL : FLAG_AGG, count),
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

    // 4. both key arrays -> LDS at the tile-local sorted position, side by side (one staging round for the 6 key bytes:
    //    with the other array as a third round of its own -- load, stage, barrier, store -- the passes ran at 3.8 TB/s, the last
    //    pass, which always staged both, at 5.3)
    u32 pos[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        pos[j] = wh[rd[j] >> 16] + (rd[j] & 0xFFFFu);
        if (FULL || p < tile_n) {
            if (ON16) { s_ext[pos[j]] = (u16)key[j]; s_keys[pos[j]] = k32in[p]; }
            else { s_keys[pos[j]] = key[j]; s_ext[pos[j]] = k16in[p]; }
        }
    }
    __syncthreads();

    // 5. look-back inside the bucket; the other lanes clear the next-digit histogram (reuses s_whist)
    // (the histogram is kept in WAVES lane-selected copies: on word / name text the digits above the one being sorted are
    //  heavily skewed, and 64 lanes of a wave adding to ONE LDS word serialise)
    const bool has_next = !LAST;
    if (has_next) for (int i = tid; i < WAVES * RADIX; i += BLOCK) s_whist[i] = 0;
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
            const u32 k32 = s_keys[p], k16 = s_ext[p];
            gidx[k] = s_gdelta[((ON16 ? k16 : k32) >> a.shift) & a.mask] + p;
            if (LAST) {
                a.keys_out64[gidx[k]] = ((u64)bucket << 56) | ((u64)k32 << (a.begin_bit + 16)) | ((u64)k16 << a.begin_bit);
            } else {
                a.k32_out[gidx[k]] = k32;
                a.k16_out[gidx[k]] = (u16)k16;
                atomicAdd(&s_whist[(u32)(lane & (WAVES - 1)) * RADIX + (((a.next_on16 ? k16 : k32) >> a.next_shift) & a.next_mask)], 1u);
            }
        }
    }
    sync_lds();   // LDS atomics above; every read of s_keys / s_ext is done
    if (has_next) {
        for (int i = tid; i < RADIX; i += BLOCK) {
            u32 v = 0;
#pragma unroll
            for (int c = 0; c < WAVES; ++c) v += s_whist[c * RADIX + i];
            if (v) atomicAdd(&a.next_hist[bucket * RADIX + i], v);
        }
    }
    // 8. the values
    u32 val[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 p = woff + j * WAVE;
        val[j] = (FULL || p < tile_n) ? vin[p] : 0u;
    }
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

template <int BLOCK, int ITEMS, bool ON16, bool LAST>
__global__ __launch_bounds__(BLOCK, ITEMS <= 14 ? 6 : 4) void seg48_onesweep_kernel(Seg48Args a) {
    constexpr int WAVES = BLOCK / WAVE;
    constexpr u32 TILE = BLOCK * ITEMS;
    __shared__ __attribute__((aligned(16))) u32 s_keys[TILE];   // reused for the other key array and the values
    __shared__ u16 s_ext[TILE];
    __shared__ u32 s_whist[WAVES * RADIX];
    __shared__ u32 s_gdelta[RADIX];
    __shared__ u32 s_wsum[RADIX / WAVE];
    __shared__ u32 s_flat;
    // the plan (bucket starts, tile prefixes, chunk starts) is only read before the tile's first barrier: it lives in s_ext,
    // which is first written after that barrier -- 2 KB that decide between two and three workgroups per CU at 14 records per thread
    u32* s_t = reinterpret_cast<u32*>(s_ext);
    u32* s_b = s_t + (RADIX + 1);
    u32* s_c = s_b + (RADIX + 1);
    static_assert((2 * (RADIX + 1) + NCHUNK + 1) * 4 <= TILE * 2, "plan fits the extension array");

    const int tid = threadIdx.x;
    u32 home = 0, t_home = 0;
    if (tid == 0) {
        home = xcc_id();
        t_home = atomicAdd(&a.ticket[home], 1u);
    }
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
        seg48_tile<true, BLOCK, ITEMS, ON16, LAST>(a, flat, first_flat, bucket, start, TILE, s_keys, s_ext, s_whist, s_gdelta, s_wsum);
    else
        seg48_tile<false, BLOCK, ITEMS, ON16, LAST>(a, flat, first_flat, bucket, start, rest, s_keys, s_ext, s_whist, s_gdelta, s_wsum);
}

// keys of 41 .. 56 bits, a text-sourced top-digit pass, enough records for the flat tile count to fit the status array
inline bool narrow48_applies(const RadixWorkspace& ws, u64 n, int begin_bit, int b, int k0) {
    return ws.block == 512 && begin_bit >= 8 && begin_bit < 24 && n >= (1u << 22) && text_pass_applies(b, k0) &&
           n / (512u * SEG48_LAST_ITEMS) + RADIX + 1 <= ws.max_tiles;
}

// Sort n records (key of every text position, position) by key bits [begin_bit, 64), stable; keys = the first src->k0
// characters (radix_prepare() + narrow_text_histogram() before).  keysA / keysB: 8 n + 64 bytes each (k32 in the first
// 4 n bytes of a buffer, k16 behind it); result: *keys_res = u64 full keys, *vals_res.
inline int radix_sort_narrow48(RadixWorkspace& ws, NarrowWorkspace& nw, hipStream_t stream, u64* keysA, u32* valsA,
                               u64* keysB, u32* valsB, u32 n, int begin_bit, u64** keys_res, u32** vals_res, const TextSource& src,
                               int64_t* vals_res64 = nullptr) {
    int rc;
    const int rem = 56 - begin_bit;                              // 33 .. 48 bits below the top digit
    const int hi_bits = rem - 16;                                // in k32: 17 .. 32
    const int np_hi = (hi_bits + RADIX_BITS - 1) / RADIX_BITS;   // 3 or 4
    const int np = 2 + np_hi;
    const SortGeom g = make_geom(n, TEXT_TILE);
    SA_HIP_CHECK(hipMemsetAsync(nw.hist, 0, NarrowWorkspace::hist_bytes(), stream));
    SA_HIP_CHECK(hipMemsetAsync(nw.tickets, 0, (size_t)NARROW_MAX_PASSES * NCHUNK * sizeof(u32), stream));
    const size_t off16 = ((size_t)n * 4 + 63) & ~(size_t)63;    // k16 of a buffer starts here
    auto k32_of = [](u64* buf) { return reinterpret_cast<u32*>(buf); };
    auto k16_of = [&](u64* buf) { return reinterpret_cast<u16*>(reinterpret_cast<u8*>(buf) + off16); };

    // top digit straight from the text: k32, k16 and the positions leave in top-digit order
    {
        if (++ws.epoch >= (1u << 30)) {
            SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
            ws.epoch = 1;
        }
        hipLaunchKernelGGL(radix_scan_hist_kernel, dim3(1), dim3(256), 0, stream, ws.hist(0), ws.base());
        TextPassArgs t;
        t.text = src.text; t.map = nw.map_dev; t.n = n; t.b = src.b; t.k0 = src.k0; t.begin_bit = begin_bit;
        t.keys_out32 = k32_of(keysB); t.ext_out16 = k16_of(keysB); t.vals_out = valsB; t.g = g; t.digit_base = ws.base();
        t.status = ws.status; t.ticket = ws.tickets(); t.epoch = ws.epoch; t.dstat = ws.dstat; t.incl_mask = SA_INCL_MASK; t.cursor = nullptr; t.cursor_stride = 1;
        if ((rc = ws.timer.start(stream, 1))) return rc;
        hipLaunchKernelGGL((text_top_pass_kernel<512, true>), dim3(g.tiles), dim3(512), 0, stream, t);
        if ((rc = ws.timer.stop(stream, (u64)n * 11u))) return rc;
        ws.pass_records += n; ws.pass_bytes += (u64)n * 11u; ws.passes += 1;
    }
    const u32 tile_mid = 512u * SEG48_ITEMS;
    hipLaunchKernelGGL(seg_plan_kernel, dim3(1), dim3(256), 0, stream, ws.hist(0), tile_mid, nw.plan);
    const u32 flat_mid = n / tile_mid + 1 + RADIX, flat_last = flat_mid;
    {   // histogram of the first narrow digit (low 8 bits of k16) per bucket
        const u32 tpb = 4;
        hipLaunchKernelGGL((seg_hist_kernel<512, SEG48_ITEMS, u16>), dim3(div_up(flat_mid, tpb)), dim3(512), 0, stream,
                           (const u16*)k16_of(keysB), nw.plan, 0, 255u, nw.hist, tpb);
    }
    u64* bin = keysB; u32* vin = valsB;
    u64* bout = keysA; u32* vout = valsA;
    for (int p = 0; p < np; ++p) {
        if (++ws.epoch >= (1u << 30)) {
            SA_HIP_CHECK(hipMemsetAsync(ws.status, 0, (size_t)ws.max_tiles * RADIX * sizeof(u64), stream));
            ws.epoch = 1;
        }
        const bool last = (p == np - 1);
        const bool on16 = p < 2;
        auto bits_of = [&](int q) { return q < 2 ? RADIX_BITS : ((q == np - 1) ? hi_bits - RADIX_BITS * (np_hi - 1) : RADIX_BITS); };
        auto shift_of = [&](int q) { return q < 2 ? RADIX_BITS * q : RADIX_BITS * (q - 2); };
        hipLaunchKernelGGL(seg_scan_kernel, dim3(RADIX), dim3(256), 0, stream, nw.hist + (size_t)p * RADIX * RADIX, nw.plan, nw.base);
        Seg48Args a;
        a.k32_in = k32_of(bin); a.k16_in = k16_of(bin); a.vals_in = vin;
        a.k32_out = k32_of(bout); a.k16_out = k16_of(bout); a.vals_out = vout;
        a.keys_out64 = last ? bout : nullptr;
        a.vals_out64 = last ? vals_res64 : nullptr;
        a.plan = nw.plan;
        a.shift = shift_of(p); a.mask = (1u << bits_of(p)) - 1u;
        a.next_shift = last ? -1 : shift_of(p + 1);
        a.next_mask = last ? 0u : ((1u << bits_of(p + 1)) - 1u);
        a.next_on16 = (p + 1 < 2) ? 1 : 0;
        a.digit_base = nw.base;
        a.next_hist = last ? nullptr : nw.hist + (size_t)(p + 1) * RADIX * RADIX;
        a.status = ws.status; a.ticket = nw.tickets + p * NCHUNK; a.epoch = ws.epoch; a.dstat = ws.dstat;
        a.begin_bit = begin_bit; a.incl_mask = SA_INCL_MASK;
        if ((rc = ws.timer.start(stream, last ? 3 : 2))) return rc;
        if (last) hipLaunchKernelGGL((seg48_onesweep_kernel<512, SEG48_LAST_ITEMS, false, true>), dim3(flat_last), dim3(512), 0, stream, a);
        else if (on16) hipLaunchKernelGGL((seg48_onesweep_kernel<512, SEG48_ITEMS, true, false>), dim3(flat_mid), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((seg48_onesweep_kernel<512, SEG48_ITEMS, false, false>), dim3(flat_mid), dim3(512), 0, stream, a);
        const u64 pass_bytes = (u64)n * (last ? 22u : 20u) + ((last && vals_res64) ? (u64)n * 8u : 0u);
        if ((rc = ws.timer.stop(stream, pass_bytes))) return rc;
        ws.pass_records += n; ws.pass_bytes += pass_bytes; ws.passes += 1;
        u64* tb = bin; bin = bout; bout = tb;
        u32* tv = vin; vin = vout; vout = tv;
    }
    SA_HIP_CHECK(hipGetLastError());
    *keys_res = bin;
    *vals_res = vin;
    return 0;
}

}  // namespace sa
