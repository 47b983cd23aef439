// sa_build.hpp -- device suffix-array construction for gfx950: alphabet compaction, packed
// initial keys, one-sweep radix sort (narrow 8-byte records for keys of <= 40 bits, 12-byte records
// otherwise; the first pass of either reads the text itself), then refinement over the ACTIVE set
// only (suffixes whose group is not yet a singleton): a direct-comparison finisher for tiny groups
// when the active set is sparse, the in-LDS group finisher (group_finish.hpp: every group that fits a
// tile is refined to the end by one workgroup), and for what is left "chunk" rounds that append the
// next characters of the text to the key and "doubling" rounds (Larsson-Sadakane with discarding)
// that append the rank of suffix i+h.  A 64-bit build also leaves the result as int64 (libsais64
// layout): stored by the sort's last pass, patched where refinement wrote later.
// tests/pipeline_model.py is the executable specification of the host logic of the rounds.
//
// Replaces, by function only, libsais / libsais64 (libsais.c:6618, libsais64.c:6657: T -> SA)
// and construct_truncated_suffix_array (engine.c:837-866).  Order: unsigned bytes, a suffix
// that is a proper prefix of another sorts first (libsais.c:703-707); truncated mode orders by
// the first L bytes with ties in text order.
//
// HBM layout (n = text bytes, M = active records of a round):
//   text     u8 [n + TEXT_PAD]   zero padded, read coalesced (keygen) or by random 8..64 B windows
//   keys0/1  u64[n] or u32[n]     ping-pong sort keys (initial sort; sized once the plan is known)
//   vals0/1  u32[n]              ping-pong suffix indices; the result buffer IS the suffix array
//   flags    u8 [n]              bit0 = group head at this SA slot
//   isa      u32[n]              rank of every suffix (allocated on the first doubling round)
//   round pool (37 B x M0)       apos/aidx/gid lists, round keys + indices ping-pong, local flags
#pragma once
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "radix_narrow.hpp"
#include "radix_narrow48.hpp"
#include "round_sort.hpp"
#include "group_finish.hpp"
#include "period_finish.hpp"

namespace sa {

constexpr int TEXT_PAD = 256;        // readable zero bytes after the text
constexpr int BLD_BLOCK = 256;
constexpr int BLD_ITEMS = 16;
constexpr int BLD_TILE = BLD_BLOCK * BLD_ITEMS;  // 4096 elements per workgroup
constexpr u32 NONE32 = 0xFFFFFFFFu;


// ---- byte histogram (libsais `freq`, libsais.c:1363-1371) -------------------------------------
// Small alphabets put many lanes of a wave on the same counter (LDS atomics to one address serialise),
// so the histogram is kept in HIST_COPIES copies selected by the lane: 27 symbols x 8 copies leave
// about one lane per address and instruction.
constexpr int HIST_COPIES = 8;
__global__ __launch_bounds__(256) void byte_hist_kernel(const u8* __restrict__ text, u64 n, u64* __restrict__ hist) {
    constexpr int CS = 257;   // copy stride: the same symbol in different copies falls into different banks
    __shared__ u32 s_h[HIST_COPIES * CS];
    for (int i = threadIdx.x; i < HIST_COPIES * CS; i += 256) s_h[i] = 0;
    __syncthreads();
    u32* my = s_h + (threadIdx.x & (HIST_COPIES - 1)) * CS;
    const u64 nvec = n / 16;
    const uint4* v = reinterpret_cast<const uint4*>(text);
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const uint4 x = v[i];
        const u32 w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&my[w[k] & 255u], 1u);
            atomicAdd(&my[(w[k] >> 8) & 255u], 1u);
            atomicAdd(&my[(w[k] >> 16) & 255u], 1u);
            atomicAdd(&my[w[k] >> 24], 1u);
        }
    }
    if (blockIdx.x == 0) {
        for (u64 i = nvec * 16 + threadIdx.x; i < n; i += blockDim.x) atomicAdd(&my[text[i]], 1u);
    }
    sync_lds();   // LDS atomics above (see sync_lds)
    u32 c = 0;
#pragma unroll
    for (int k = 0; k < HIST_COPIES; ++k) c += s_h[k * CS + threadIdx.x];
    if (c) atomicAdd((unsigned long long*)&hist[threadIdx.x], (unsigned long long)c);
}

// ---- pilot: how repetitive is the text at the key length the i.i.d. estimate would choose? -------------
// PILOT_SAMPLES evenly spaced windows of `w` raw bytes are inserted into a hash set (linear probing,
// 64-bit CAS); *dups counts the windows that were already present.  On an i.i.d. text practically none
// are; on word / name / log-like texts a large share is, and the initial sort then takes the longest
// key that fits instead (fewer, smaller refinement rounds).
constexpr u32 PILOT_SAMPLES = 1u << 17;
constexpr u32 PILOT_SLOTS = 1u << 20;
__global__ __launch_bounds__(256) void pilot_kernel(const u8* __restrict__ text, u64 n, int w, unsigned long long* __restrict__ table,
                                                    u32* __restrict__ dups) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= PILOT_SAMPLES) return;
    const u64 p = (n - 8) / PILOT_SAMPLES * s;          // n >= 16 * PILOT_SAMPLES
    u64 v;
    __builtin_memcpy(&v, text + p, 8);
    if (w < 8) v &= (1ull << (8 * w)) - 1ull;
    const unsigned long long key = v * 0x9E3779B97F4A7C15ull | 1ull;   // odd multiplier: injective; never 0
    u32 slot = (u32)(key >> 40) & (PILOT_SLOTS - 1);
    for (int probe = 0; probe < 16; ++probe) {
        const unsigned long long old = atomicCAS(&table[slot], 0ull, key);
        if (old == 0ull) return;
        if (old == key) { atomicAdd(dups, 1u); return; }
        slot = (slot + 1) & (PILOT_SLOTS - 1);
    }
}

// ---- initial keys: key[i] = codes of T[i..i+k0) packed MSB-first, b bits each ----------------------
// One workgroup stages BLD_TILE + k0 text bytes as codes in LDS (16-byte global loads), every
// thread then assembles the keys of 16 positions (stride 256 -> 8-byte coalesced stores).
// While the keys are in registers the workgroup also accumulates the per-chunk histogram of the
// first digit of the sort that follows (ghist != nullptr), which saves an 8-byte/record pre-pass
// (later passes get their histograms from the pass before them, radix_sort.hpp).
__global__ __launch_bounds__(BLD_BLOCK) void keygen_kernel(const u8* __restrict__ text, u64 n, CodeMap map, int b,
                                                           int k0, u64* __restrict__ keys, SortGeom g, int shift0,
                                                           u32 mask0, u32* __restrict__ ghist) {
    __shared__ u16 s_map[256];
    __shared__ u16 s_codes[BLD_TILE + 64 + 16];
    __shared__ u32 s_h[RADIX];
    s_map[threadIdx.x] = map.code[threadIdx.x];
    s_h[threadIdx.x] = 0;
    const u64 ntiles = (n + BLD_TILE - 1) / BLD_TILE;
    const u64 per = (ntiles + gridDim.x - 1) / gridDim.x;
    const u64 t_lo = (u64)blockIdx.x * per;
    const u64 t_hi = (t_lo + per < ntiles) ? t_lo + per : ntiles;
    u32 cur_chunk = chunk_of_tile((u32)((t_lo * BLD_TILE) >> g.tile_shift), g.tpc);
    for (u64 tile = t_lo; tile < t_hi; ++tile) {
        const u64 base = tile * BLD_TILE;
        if (ghist) {
            const u32 c = chunk_of_tile((u32)(base >> g.tile_shift), g.tpc);
            if (c != cur_chunk) { hist_flush(s_h, ghist, cur_chunk); cur_chunk = c; }
        }
        sync_lds();   // previous tile's LDS atomics + code reads (see sync_lds)
        {   // body: 16 bytes per thread (text is padded by >= 16 readable bytes past n)
            uint4 x = make_uint4(0, 0, 0, 0);
            if (base + (u64)threadIdx.x * 16 < n) x = *reinterpret_cast<const uint4*>(text + base + (u64)threadIdx.x * 16);
            const u32 w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const u64 p = base + (u64)threadIdx.x * 16 + k;
                const u32 byte = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
                s_codes[threadIdx.x * 16 + k] = (p < n) ? s_map[byte] : (u16)0;
            }
        }
        if (threadIdx.x < 64 + 16) {  // halo
            const u64 p = base + BLD_TILE + threadIdx.x;
            s_codes[BLD_TILE + threadIdx.x] = (p < n) ? s_map[text[p]] : (u16)0;
        }
        sync_lds();
#pragma unroll 4
        for (int it = 0; it < BLD_ITEMS; ++it) {
            const u32 l = it * BLD_BLOCK + threadIdx.x;
            const u64 p = base + l;
            if (p < n) {
                u64 key = 0;
                int sh = 64;
                for (int j = 0; j < k0; ++j) {
                    sh -= b;
                    key |= (u64)s_codes[l + j] << sh;
                }
                keys[p] = key;
                if (ghist) atomicAdd(&s_h[(u32)(key >> shift0) & mask0], 1u);
            }
        }
    }
    if (ghist && t_lo < t_hi) hist_flush(s_h, ghist, cur_chunk);
}

// ---- head / active flags + per-tile counts -------------------------------------------------------------
// lf[j]: bit0 = head (key differs from predecessor), bit1 = active (group of j has > 1 member).
// Optional write-back of a refinement round: SA[apos[j]] = sidx[j]; gflags[apos[j]] |= head.
// counts[tile] = {#active, #active heads} of the tile.
// Optional bucket directory of the query path (first flags pass of a build only, dirargs.dir != nullptr):
// dir[bkt] = first slot whose key has top-dbits >= bkt.  Slot j owns the buckets (top(K[j-1]), top(K[j])];
// runs of up to DIR_INLINE buckets are written here, longer ones (unused codes of the compacted
// alphabet leave holes of up to 2^dbits / 8 buckets) are queued for dir_fill_kernel.
// The directory is written early in the build; reading it once at the end leaves its 2^dbits * 4 bytes
// (268 MB at the default 26 bits for n = 1e9, 4 MB at n = 1e7) in the memory-side cache for the first query batch
// as far as they fit, as a directory built last would be.
__global__ __launch_bounds__(256) void dir_touch_kernel(const uint4* __restrict__ dir16, u64 n16, u32* __restrict__ sink) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = dir16[i];
        acc |= v.x & v.y & v.z & v.w;
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;   // slots are < n <= 2^32 - 2: never taken, keeps the loads alive
}

__global__ __launch_bounds__(256) void dir_fill_kernel(DirArgs d) {
    const u32 count = *d.gap_count < d.gap_cap ? *d.gap_count : d.gap_cap;
    for (u32 e = blockIdx.x; e < count; e += gridDim.x) {
        const uint4 g = d.gaps[e];
        for (u64 bkt = (u64)g.x + threadIdx.x; bkt <= (u64)g.y; bkt += blockDim.x) d.dir[bkt] = g.z;
    }
}

// NARROW: the keys are the u32 narrow keys a narrow-record sort leaves behind (radix_narrow.hpp, keep_narrow): the full
// key of slot j is (bucket(j) << 56) | (narrow[j] << lo_shift), bucket(j) = the b with bstart[b] <= j < bstart[b+1].
// The kernel rebuilds the u64 keys in registers (4 instead of 8 bytes read per slot, and the sort's last pass has
// 8 bytes per slot less to write); everything after the loads is the same code.
struct NarrowKeys {
    const u32* keys32;
    const u32* bstart;   // [257], bstart[256] = n
    int lo_shift;
};
template <bool NARROW>
__global__ __launch_bounds__(BLD_BLOCK) void flags_kernel(const u64* __restrict__ keys, NarrowKeys nk, u32 n, u8* __restrict__ lf,
                                                          uint2* __restrict__ counts, const u32* __restrict__ apos,
                                                          const u32* __restrict__ sidx, u32* __restrict__ sa_out,
                                                          u8* __restrict__ gflags, DirArgs dirargs) {
    __shared__ u32 s_a[BLD_BLOCK / WAVE], s_h[BLD_BLOCK / WAVE];
    __shared__ u32 s_b[NARROW ? 257 : 1];
    if (NARROW) {
        for (int i = threadIdx.x; i <= 256; i += BLD_BLOCK) s_b[i] = nk.bstart[i];
        __syncthreads();
    }
    const u32 ntiles = (u32)(((u64)n + BLD_TILE - 1) / BLD_TILE);
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // SA_FLAGS_PERSIST: a resident grid walks the tiles
    const u64 base = (u64)tile * BLD_TILE;
    u32 ca = 0, ch = 0;
    u32 bkt = 0;   // NARROW: bucket of the slot looked at last; a thread's slots only move forward
    if (NARROW) {
        // bucket of the first slot this thread looks at: the last b with bstart[b] <= slot (empty buckets share their
        // successor's start, so this is the non-empty one that holds the slot); later slots walk on from there
        const u64 j0 = base + (u64)threadIdx.x * 4;
        const u64 jf = (j0 > 0) ? j0 - 1 : 0;
        u32 lo = 0, hi = 256;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const u32 mid = (lo + hi) >> 1;
            if ((u64)s_b[mid] <= jf) lo = mid; else hi = mid;
        }
        bkt = lo;
    }
    // four consecutive elements per thread and step: two 16-byte key loads + the two neighbouring keys instead of
    // three 8-byte loads per element, and one 4-byte store of the four flag bytes instead of four 1-byte stores
    // (the key and flag buffers are 16-byte aligned; the flag buffer carries >= 64 bytes of slack)
    static_assert(BLD_ITEMS % 4 == 0, "four elements per step");
#pragma unroll
    for (int it = 0; it < BLD_ITEMS / 4; ++it) {
        const u64 j0 = base + (u64)it * (BLD_BLOCK * 4) + (u64)threadIdx.x * 4;
        if (j0 >= n) continue;
        u64 k[6];   // keys j0-1 .. j0+4
        if (NARROW) {
            u32 w[6];
            if (j0 + 4 <= n) {
                const uint4 x = *reinterpret_cast<const uint4*>(nk.keys32 + j0);
                w[1] = x.x; w[2] = x.y; w[3] = x.z; w[4] = x.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) w[1 + e] = (j0 + e < n) ? nk.keys32[j0 + e] : 0u;
            }
            w[0] = (j0 > 0) ? nk.keys32[j0 - 1] : 0u;
            w[5] = (j0 + 4 < n) ? nk.keys32[j0 + 4] : 0u;
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const u64 j = j0 + e - 1;   // e == 0 at j0 == 0 is never used
                if (j < n && !(e == 0 && j0 == 0)) {
                    while ((u64)s_b[bkt + 1] <= j) ++bkt;   // bstart[256] = n > j
                    k[e] = ((u64)bkt << 56) | ((u64)w[e] << nk.lo_shift);
                } else k[e] = 0ull;
            }
        } else {
        if (j0 + 4 <= n) {
            const uint4 x0 = *reinterpret_cast<const uint4*>(keys + j0);
            const uint4 x1 = *reinterpret_cast<const uint4*>(keys + j0 + 2);
            k[1] = ((u64)x0.y << 32) | x0.x; k[2] = ((u64)x0.w << 32) | x0.z;
            k[3] = ((u64)x1.y << 32) | x1.x; k[4] = ((u64)x1.w << 32) | x1.z;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) k[1 + e] = (j0 + e < n) ? keys[j0 + e] : 0ull;
        }
        k[0] = (j0 > 0) ? keys[j0 - 1] : 0ull;
        k[5] = (j0 + 4 < n) ? keys[j0 + 4] : 0ull;
        }
        u32 fl = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const u64 j = j0 + e;
            if (j < n) {
                const u64 kk = k[1 + e], kprev = k[e];
                const bool head = (j == 0) || (kprev != kk);
                const bool next_head = (j + 1 == n) || (k[2 + e] != kk);
                if (dirargs.dir && head) {
                    const int ds = 64 - dirargs.dbits;
                    const u32 bj = (u32)(kk >> ds);
                    const u32 first = (j == 0) ? 0u : (u32)(kprev >> ds) + 1u;
                    if (first <= bj) dir_emit(dirargs, first, bj, (u32)j);
                }
                if (dirargs.dir && j + 1 == n)   // buckets above the last key, and the end marker dir[2^dbits]
                    dir_emit(dirargs, (u32)(kk >> (64 - dirargs.dbits)) + 1u, 1u << dirargs.dbits, n);
                const bool act = !(head && next_head);
                fl |= (u32)((head ? 1 : 0) | (act ? 2 : 0)) << (8 * e);
                ca += act;
                ch += (act && head);
                if (apos) {
                    const u32 slot = apos[j];
                    sa_out[slot] = sidx[j];
                    if (head) gflags[slot] = 1;
                }
            }
        }
        if (j0 + 4 <= n) *reinterpret_cast<u32*>(lf + j0) = fl;
        else for (int e = 0; e < 4 && j0 + e < n; ++e) lf[j0 + e] = (u8)(fl >> (8 * e));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ca += __shfl_down(ca, o);
        ch += __shfl_down(ch, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_a[wave] = ca; s_h[wave] = ch; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 ta = 0, th = 0;
        for (int w = 0; w < BLD_BLOCK / WAVE; ++w) { ta += s_a[w]; th += s_h[w]; }
        counts[tile] = make_uint2(ta, th);
    }
    __syncthreads();   // s_a / s_h are reused by the next tile
    }
}

// ---- the first flags pass of a near-random text: active records straight to a per-tile staging area -----------------
// On the narrow-record plan (keys of <= 40 bits: the text is near-random and a fraction of a percent of the suffixes is still
// tied) the full pass above writes a flag byte per slot (n bytes) that compact_kernel reads back (n bytes) to find those few
// records.  Here the pass keeps its flags in registers: head / active bits of a thread's 16 slots as two 16-bit masks, the
// threads that hold active slots reserve places in the tile's staging row (LITE_CAP entries of {slot, suffix, head bit}) by
// one atomic each, and lite_gather_kernel moves the rows, in slot order, to the dense lists once the per-tile counts have been scanned.  No flag array at
// all: the later phases that mark heads in it (refinement write-backs, doubling) get it from materialise_flags() first --
// on this plan the tiny-group finisher normally resolves everything and nobody asks.  A tile with more than LITE_CAP
// active slots (6.25 %: the tiny-group finisher's own limit is M * 16 <= n) raises *overflow and the build repeats the
// pass in its full form.  The bucket directory is written exactly as above.
__global__ __launch_bounds__(BLD_BLOCK, 8) void flags_lite_kernel(NarrowKeys nk, u32 n, uint2* __restrict__ counts, DirArgs dirargs, LiteArgs o) {
    constexpr int STEPS = BLD_ITEMS / 4;
    static_assert(STEPS == 4, "sixteen slots per thread");
    __shared__ u32 s_b[257];
    for (int i = threadIdx.x; i <= 256; i += BLD_BLOCK) s_b[i] = nk.bstart[i];
    __syncthreads();
    const u32 ntiles = (u32)(((u64)n + BLD_TILE - 1) / BLD_TILE);
    // Round 4, second form: no workgroup scan and no barrier per tile.  A thread that holds active slots (5 % of the threads on
    // D1) reserves their places in the tile's staging row with ONE 64-bit atomic on the tile's {active, heads} pair (zeroed by
    // the host) and writes them there, in slot order among themselves; lite_gather_kernel puts a row into slot order (a
    // rank by counting over its dozen entries).  The first form scanned the threads' counts per tile (two barriers, 2.04 ms per
    // build under rocprofv3 against 1.57 + 0.56 for the full pass and its compaction).
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u64 base = (u64)tile * BLD_TILE;
        u32 bkt;
        {
            const u64 j0 = base + (u64)threadIdx.x * 4;
            const u64 jf = (j0 > 0) ? j0 - 1 : 0;
            u32 lo = 0, hi = 256;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32 mid = (lo + hi) >> 1;
                if ((u64)s_b[mid] <= jf) lo = mid; else hi = mid;
            }
            bkt = lo;
        }
        u32 amask = 0, hmask = 0;   // bit 4 * it + e: slot base + it * 1024 + tid * 4 + e is active / a head
#pragma unroll
        for (int it = 0; it < STEPS; ++it) {
            const u64 j0 = base + (u64)it * (BLD_BLOCK * 4) + (u64)threadIdx.x * 4;
            if (j0 >= n) continue;
            u32 w[6];
            if (j0 + 4 <= n) {
                const uint4 x = *reinterpret_cast<const uint4*>(nk.keys32 + j0);
                w[1] = x.x; w[2] = x.y; w[3] = x.z; w[4] = x.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) w[1 + e] = (j0 + e < n) ? nk.keys32[j0 + e] : 0u;
            }
            w[0] = (j0 > 0) ? nk.keys32[j0 - 1] : 0u;
            w[5] = (j0 + 4 < n) ? nk.keys32[j0 + 4] : 0u;
            u64 k[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const u64 j = j0 + e - 1;
                if (j < n && !(e == 0 && j0 == 0)) {
                    while ((u64)s_b[bkt + 1] <= j) ++bkt;
                    k[e] = ((u64)bkt << 56) | ((u64)w[e] << nk.lo_shift);
                } else k[e] = 0ull;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u64 j = j0 + e;
                if (j < n) {
                    const u64 kk = k[1 + e], kprev = k[e];
                    const bool head = (j == 0) || (kprev != kk);
                    const bool next_head = (j + 1 == n) || (k[2 + e] != kk);
                    if (dirargs.dir && head) {
                        const int ds = 64 - dirargs.dbits;
                        const u32 bj = (u32)(kk >> ds);
                        const u32 first = (j == 0) ? 0u : (u32)(kprev >> ds) + 1u;
                        if (first <= bj) dir_emit(dirargs, first, bj, (u32)j);
                    }
                    if (dirargs.dir && j + 1 == n)
                        dir_emit(dirargs, (u32)(kk >> (64 - dirargs.dbits)) + 1u, 1u << dirargs.dbits, n);
                    if (!(head && next_head)) {
                        amask |= 1u << (4 * it + e);
                        if (head) hmask |= 1u << (4 * it + e);
                    }
                }
            }
        }
        if (amask) {
            const u32 c = (u32)__popc(amask), ch = (u32)__popc(hmask);
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(&counts[tile]), (unsigned long long)c | ((unsigned long long)ch << 32));
            u32 r = (u32)old;
            if (r + c > LITE_CAP) __hip_atomic_store(o.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {
#pragma unroll
                for (int b2 = 0; b2 < 16; ++b2) {
                    if (amask & (1u << b2)) {
                        const u64 j = base + (u64)(b2 >> 2) * (BLD_BLOCK * 4) + (u64)threadIdx.x * 4 + (u32)(b2 & 3);
                        const u64 at = (u64)tile * LITE_CAP + r;
                        o.st_pos[at] = (u32)j;
                        o.st_idx[at] = o.sa[j];
                        o.st_head[at] = (hmask >> b2) & 1u;
                        ++r;
                    }
                }
            }
        }
    }
}

// staging rows -> the dense active lists (slot, suffix, dense group id), one wave per tile; offsets = the scanned counts.  A row
// arrives in the order its threads reserved their places: it is put into slot order first (rank by counting in LDS)
__global__ __launch_bounds__(256) void lite_gather_kernel(const uint2* __restrict__ offsets, u32 ntiles, const u32* __restrict__ totals,
                                                          LiteArgs o, u32* __restrict__ dst_pos, u32* __restrict__ dst_idx,
                                                          u32* __restrict__ dst_gid) {
    __shared__ u32 s_pos[4][LITE_CAP], s_idx[4][LITE_CAP];
    __shared__ u8 s_head[4][LITE_CAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 tile = blockIdx.x * (blockDim.x >> 6) + (u32)wv;
    if (tile >= ntiles) return;
    const uint2 off = offsets[tile];
    const u32 end = (tile + 1 < ntiles) ? offsets[tile + 1].x : totals[0];
    const u32 cnt = end - off.x;   // <= LITE_CAP
    if (cnt == 0) return;
    u32 pos[LITE_CAP / 64], idx[LITE_CAP / 64];
    u8 hd[LITE_CAP / 64];
#pragma unroll
    for (int e = 0; e < (int)(LITE_CAP / 64); ++e) {
        const u32 r = (u32)e * 64 + lane;
        const u64 at = (u64)tile * LITE_CAP + (r < cnt ? r : 0);
        pos[e] = o.st_pos[at]; idx[e] = o.st_idx[at]; hd[e] = o.st_head[at];
        if (r < cnt) s_pos[wv][r] = pos[e];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < (int)(LITE_CAP / 64); ++e) {
        const u32 r = (u32)e * 64 + lane;
        if (r < cnt) {
            u32 rank = 0;
            for (u32 k = 0; k < cnt; ++k) rank += (s_pos[wv][k] < pos[e]) ? 1u : 0u;   // slots are distinct
            s_idx[wv][rank] = idx[e];
            s_head[wv][rank] = hd[e];
            idx[e] = rank;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < (int)(LITE_CAP / 64); ++e) {   // (every lane has read s_pos: the sorted slots may take its place)
        const u32 r = (u32)e * 64 + lane;
        if (r < cnt) s_pos[wv][idx[e]] = pos[e];
    }
    __builtin_amdgcn_wave_barrier();
    u32 heads = off.y;
    const u64 lt = lanemask_lt();
    for (u32 r0 = 0; r0 < cnt; r0 += 64) {
        const u32 r = r0 + lane;
        const bool in = r < cnt;
        const bool head = in && s_head[wv][in ? r : 0] != 0;
        const u64 bh = __ballot(head);
        if (in) {
            const u32 m = off.x + r;
            dst_pos[m] = s_pos[wv][r];
            dst_idx[m] = s_idx[wv][r];
            dst_gid[m] = heads + (u32)__popcll(bh & lt) + (head ? 1u : 0u) - 1u;
        }
        heads += (u32)__popcll(bh);
    }
}

// flags[slot] = 1 for every list element that a finisher has resolved (its slot starts a group of one): what the finisher
// itself would have stored had the flag array existed then
__global__ void mark_done_heads_kernel(const u32* __restrict__ apos, const u8* __restrict__ done, u32 m, u8* __restrict__ gflags) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        if (done[i]) gflags[apos[i]] = 1;
}

// exclusive scan of the per-tile {active, heads} pairs in three coalesced steps:
//   reduce 1024 tiles per workgroup -> scan the partials (one workgroup) -> scan inside each group.
__device__ __forceinline__ uint2 block_excl_scan_1024(uint2 v, u32* s_a, u32* s_h, uint2* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 ia = v.x, ih = v.y;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 ta = __shfl_up(ia, o), th = __shfl_up(ih, o);
        if (lane >= o) { ia += ta; ih += th; }
    }
    if (lane == 63) { s_a[wave] = ia; s_h[wave] = ih; }
    __syncthreads();
    u32 oa = 0, oh = 0, ta = 0, th = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) { oa += s_a[w]; oh += s_h[w]; }
        ta += s_a[w]; th += s_h[w];
    }
    __syncthreads();
    if (total) *total = make_uint2(ta, th);
    return make_uint2(oa + ia - v.x, oh + ih - v.y);
}

__global__ __launch_bounds__(1024) void counts_reduce_kernel(const uint2* __restrict__ counts, u32 ntiles,
                                                             uint2* __restrict__ partial) {
    __shared__ u32 s_a[16], s_h[16];
    const u32 i = blockIdx.x * 1024 + threadIdx.x;
    uint2 v = (i < ntiles) ? counts[i] : make_uint2(0, 0);
    uint2 tot;
    (void)block_excl_scan_1024(v, s_a, s_h, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// nparts <= 1024 * PER; one workgroup; partial[] becomes exclusive, totals[0..1] = sums
__global__ __launch_bounds__(1024) void counts_scan_partials_kernel(uint2* __restrict__ partial, u32 nparts,
                                                                    u32* __restrict__ totals) {
    __shared__ u32 s_a[16], s_h[16];
    const u32 per = (nparts + 1023) / 1024;
    const u32 lo = threadIdx.x * per;
    const u32 hi = (lo + per < nparts) ? lo + per : nparts;
    uint2 v = make_uint2(0, 0);
    for (u32 i = lo; i < hi; ++i) { v.x += partial[i].x; v.y += partial[i].y; }
    uint2 tot;
    uint2 e = block_excl_scan_1024(v, s_a, s_h, &tot);
    for (u32 i = lo; i < hi; ++i) {
        const uint2 c = partial[i];
        partial[i] = e;
        e.x += c.x; e.y += c.y;
    }
    if (threadIdx.x == 0) { totals[0] = tot.x; totals[1] = tot.y; }
}

__global__ __launch_bounds__(1024) void counts_apply_kernel(uint2* __restrict__ counts, u32 ntiles,
                                                            const uint2* __restrict__ partial) {
    __shared__ u32 s_a[16], s_h[16];
    const u32 i = blockIdx.x * 1024 + threadIdx.x;
    uint2 v = (i < ntiles) ? counts[i] : make_uint2(0, 0);
    const uint2 e = block_excl_scan_1024(v, s_a, s_h, nullptr);
    const uint2 base = partial[blockIdx.x];
    if (i < ntiles) counts[i] = make_uint2(base.x + e.x, base.y + e.y);
}

// ---- compaction of the active elements ---------------------------------------------------------------
// dst_pos[m] = SA slot, dst_idx[m] = suffix index, dst_gid[m] = dense id of its group among the
// active groups.  src_pos == nullptr: the domain is the whole SA (slot = j).
__global__ __launch_bounds__(BLD_BLOCK) void compact_kernel(const u8* __restrict__ lf, u32 n,
                                                            const uint2* __restrict__ offsets,
                                                            const u32* __restrict__ src_pos,
                                                            const u32* __restrict__ src_idx, u32* __restrict__ dst_pos,
                                                            u32* __restrict__ dst_idx, u32* __restrict__ dst_gid) {
    // Each thread owns 16 consecutive elements (one 16-byte load of flags; the flag buffers carry 64
    // bytes of slack), so one workgroup scan orders the whole tile.
    constexpr int WAVES = BLD_BLOCK / WAVE;
    __shared__ u32 s_a[WAVES], s_h[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 j0 = (u64)blockIdx.x * BLD_TILE + (u64)threadIdx.x * BLD_ITEMS;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (j0 < n) v = *reinterpret_cast<const uint4*>(lf + j0);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    u32 amask = 0, hmask = 0;   // bit k: element j0+k is active / an active head
#pragma unroll
    for (int k = 0; k < BLD_ITEMS; ++k) {
        const u32 f = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
        const bool in = (j0 + k) < n;
        if (in && (f & 2u)) { amask |= 1u << k; if (f & 1u) hmask |= 1u << k; }
    }
    const u32 ca = (u32)__popc(amask), ch = (u32)__popc(hmask);
    u32 ia = ca, ih = ch;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 ta = __shfl_up(ia, o), th = __shfl_up(ih, o);
        if (lane >= o) { ia += ta; ih += th; }
    }
    if (lane == 63) { s_a[wave] = ia; s_h[wave] = ih; }
    __syncthreads();
    const uint2 off = offsets[blockIdx.x];
    u32 m = off.x + ia - ca, g = off.y + ih - ch;   // exclusive prefix of this thread
    for (int q = 0; q < wave; ++q) { m += s_a[q]; g += s_h[q]; }
    if (amask == 0) return;
#pragma unroll
    for (int k = 0; k < BLD_ITEMS; ++k) {
        if (amask & (1u << k)) {
            if (hmask & (1u << k)) ++g;
            const u64 j = j0 + k;
            dst_pos[m] = src_pos ? src_pos[j] : (u32)j;
            dst_idx[m] = src_idx[j];
            dst_gid[m] = g - 1u;
            ++m;
        }
    }
}

// Dense variant (a large share of the domain is active): elements are visited in lane-strided order so that the reads of
// src_pos / src_idx and the compacted stores coalesce.  A wave owns 1024 consecutive elements of the tile (16 rows of 64): its
// flag bytes are requested together, counted by ballots, and ONE workgroup exchange gives the wave its base -- the first form
// exchanged counts after every row of 256 (32 barriers per tile, every row's flag load waiting behind one).
__global__ __launch_bounds__(BLD_BLOCK) void compact_dense_kernel(const u8* __restrict__ lf, u32 n,
                                                                  const uint2* __restrict__ offsets,
                                                                  const u32* __restrict__ src_pos,
                                                                  const u32* __restrict__ src_idx, u32* __restrict__ dst_pos,
                                                                  u32* __restrict__ dst_idx, u32* __restrict__ dst_gid) {
    constexpr int WAVES = BLD_BLOCK / WAVE;
    __shared__ u32 s_a[WAVES], s_h[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 lt = lanemask_lt();
    const u64 wbase = (u64)blockIdx.x * BLD_TILE + (u64)wave * (WAVE * BLD_ITEMS);
    u32 f[BLD_ITEMS];
#pragma unroll
    for (int it = 0; it < BLD_ITEMS; ++it) {
        const u64 j = wbase + (u64)it * WAVE + lane;
        const u32 v = lf[j < n ? j : (u64)n - 1];   // (no branch around the load; n >= 1)
        f[it] = j < n ? v : 0u;
    }
    u32 ca = 0, ch = 0;
#pragma unroll
    for (int it = 0; it < BLD_ITEMS; ++it) {
        ca += (u32)__popcll(__ballot((f[it] & 2u) != 0));
        ch += (u32)__popcll(__ballot((f[it] & 3u) == 3u));
    }
    if (lane == 0) { s_a[wave] = ca; s_h[wave] = ch; }
    __syncthreads();
    const uint2 off = offsets[blockIdx.x];
    u32 a_off = off.x, h_off = off.y;
    for (int w = 0; w < wave; ++w) { a_off += s_a[w]; h_off += s_h[w]; }
#pragma unroll
    for (int it = 0; it < BLD_ITEMS; ++it) {
        const u64 j = wbase + (u64)it * WAVE + lane;
        const bool act = (f[it] & 2u) != 0;
        const bool ah = (f[it] & 3u) == 3u;
        const u64 ba = __ballot(act), bh = __ballot(ah);
        if (act) {
            const u32 m = a_off + (u32)__popcll(ba & lt);
            const u32 g = h_off + (u32)__popcll(bh & lt) + (ah ? 1u : 0u) - 1u;
            dst_pos[m] = src_pos ? src_pos[j] : (u32)j;
            dst_idx[m] = src_idx[j];
            dst_gid[m] = g;
        }
        a_off += (u32)__popcll(ba); h_off += (u32)__popcll(bh);
    }
}

// ---- round keys ------------------------------------------------------------------------------------------
// chunk round: key = gid << (64-gb) | next kc characters of the text after depth h (b bits each).
// The characters are fetched as unaligned 8-byte words (the text is zero padded, so reads past n
// stay in bounds; bytes at positions >= n are masked to code 0), one word per 8 characters,
// instead of one dependent byte load per character.
__global__ __launch_bounds__(256) void chunk_keys_kernel(const u8* __restrict__ text, u64 n, CodeMap map, int b,
                                                         const u32* __restrict__ aidx, const u32* __restrict__ gid,
                                                         u32 m_count, u32 h, int kc, int gb, u64* __restrict__ keys) {
    __shared__ u16 s_map[256];
    s_map[threadIdx.x] = map.code[threadIdx.x];
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x; m < m_count; m += stride) {
        const u64 start = (u64)aidx[m] + h;
        u64 key = gb ? ((u64)gid[m] << (64 - gb)) : 0ull;
        int sh = 64 - gb;
        // start <= n for every active suffix (members of a group share h real characters)
        const u64 avail = n - (start < n ? start : n);   // real characters from `start`
        for (int j0 = 0; j0 < kc; j0 += 8) {
            u64 w;
            __builtin_memcpy(&w, text + start + j0, 8);   // little-endian: byte i at bits [8i, 8i+8)
            const int lim = (kc - j0) < 8 ? (kc - j0) : 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < lim) {
                    sh -= b;
                    const u32 byte = (u32)(w >> (8 * j)) & 255u;
                    const u64 c = ((u64)(j0 + j) < avail) ? (u64)s_map[byte] : 0ull;
                    key |= c << sh;
                }
            }
        }
        keys[m] = key;
    }
}

// doubling round: key = gid << rb | (rank of suffix idx+h) + 1, 0 when idx+h == n
__global__ __launch_bounds__(256) void doubling_keys_kernel(const u32* __restrict__ isa, u64 n,
                                                            const u32* __restrict__ aidx, const u32* __restrict__ gid,
                                                            u32 m_count, u64 h, int rb, u64* __restrict__ keys) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x; m < m_count; m += stride) {
        const u64 p = (u64)aidx[m] + h;
        const u64 k2 = (p < n) ? (u64)isa[p] + 1ull : 0ull;
        keys[m] = ((u64)gid[m] << rb) | k2;
    }
}

// ---- tiny-group finisher ---------------------------------------------------------------------------------
// Groups of at most TINY_MAX suffixes are ordered by ONE lane with direct text comparisons (8-byte
// big-endian words from depth h) instead of going through key generation + an 8-pass sort round.  On
// near-random texts every tied group is a pair, so this replaces the whole refinement round.
// Full mode: a group is only written when all its members differ within TINY_DEPTH further bytes
// (otherwise it stays active for the rounds).  Truncated mode compares up to depth L and breaks ties by
// suffix index = text order, as every stable sort round would.
constexpr int TINY_MAX = 8;
constexpr u32 TINY_DEPTH = 64;

// <0, >0: order of suffixes a, b by their bytes [h, h+limit); 0: equal there.  A suffix that ends sorts first.
__device__ __forceinline__ int cmp_suffix_pair(const u8* __restrict__ text, u64 n, u64 a, u64 b, u64 h, u32 limit) {
    const u64 pa = a + h, pb = b + h;
    for (u32 d = 0; d < limit; d += 8) {
        const u32 want = (limit - d) < 8u ? (limit - d) : 8u;
        const u64 ra = (pa + d < n) ? n - (pa + d) : 0, rb = (pb + d < n) ? n - (pb + d) : 0;
        const u32 la = ra < want ? (u32)ra : want, lb = rb < want ? (u32)rb : want;
        u64 wa = 0, wb = 0;
        if (la) { __builtin_memcpy(&wa, text + pa + d, 8); wa = __builtin_bswap64(wa); if (la < 8) wa &= ~(~0ull >> (8 * la)); }
        if (lb) { __builtin_memcpy(&wb, text + pb + d, 8); wb = __builtin_bswap64(wb); if (lb < 8) wb &= ~(~0ull >> (8 * lb)); }
        if (wa != wb) return wa < wb ? -1 : 1;
        if (la != lb) return la < lb ? -1 : 1;
        if (la < want) return 0;   // both ended (only possible for a == b)
    }
    return 0;
}

__global__ __launch_bounds__(256) void tiny_groups_kernel(const u8* __restrict__ text, u64 n, const u32* __restrict__ aidx,
                                                          const u32* __restrict__ gid, const u32* __restrict__ apos, u32 m_count,
                                                          u64 h, u32 limit, int truncated, u32* __restrict__ sa,
                                                          u8* __restrict__ gflags, u8* __restrict__ done,
                                                          int64_t* __restrict__ sa64 = nullptr) {   // 64-bit build: the int64 copy too
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x; m < m_count; m += stride) {
        const u32 g = gid[m];
        if (m > 0 && gid[m - 1] == g) continue;   // only the first member of a group works
        int size = 1;
        while (size <= TINY_MAX && m + size < m_count && gid[m + size] == g) ++size;
        if (size > TINY_MAX) continue;
        u32 v[TINY_MAX];
#pragma unroll
        for (int k = 0; k < TINY_MAX; ++k) v[k] = (k < size) ? aidx[m + k] : 0xFFFFFFFFu;   // sentinel sorts last
        bool unresolved = false;
        // 19-comparator sorting network for 8 inputs (static indices: the array stays in registers)
#define SA_CE(i, j)                                                                                  \
        {                                                                                            \
            const u32 x = v[i], y = v[j];                                                            \
            int c;                                                                                   \
            if (y == 0xFFFFFFFFu || unresolved) c = -1;                                              \
            else if (x == 0xFFFFFFFFu) c = 1;                                                        \
            else {                                                                                   \
                c = cmp_suffix_pair(text, n, x, y, h, limit);                                        \
                if (c == 0) { if (truncated) c = x < y ? -1 : 1; else unresolved = true; }           \
            }                                                                                        \
            if (c > 0) { v[i] = y; v[j] = x; }                                                       \
        }
        SA_CE(0, 1) SA_CE(2, 3) SA_CE(4, 5) SA_CE(6, 7)
        SA_CE(0, 2) SA_CE(1, 3) SA_CE(4, 6) SA_CE(5, 7)
        SA_CE(1, 2) SA_CE(5, 6) SA_CE(0, 4) SA_CE(3, 7)
        SA_CE(1, 5) SA_CE(2, 6)
        SA_CE(1, 4) SA_CE(3, 6)
        SA_CE(2, 4) SA_CE(3, 5)
        SA_CE(3, 4)
#undef SA_CE
        // a network does not compare every adjacent pair of its output: make sure the order is strict (a pair has
        // been compared by SA_CE(0, 1) already: equal -> unresolved)
        if (!truncated && size > 2) {
#pragma unroll
            for (int k = 0; k + 1 < TINY_MAX; ++k)
                if (k + 1 < size && !unresolved && cmp_suffix_pair(text, n, v[k], v[k + 1], h, limit) >= 0) unresolved = true;
        }
        if (unresolved) continue;
#pragma unroll
        for (int k = 0; k < TINY_MAX; ++k) {
            if (k < size) {
                const u32 slot = apos[m + k];
                sa[slot] = v[k];
                if (sa64) sa64[slot] = (int64_t)v[k];   // (uniform)
                if (gflags) gflags[slot] = 1;   // (uniform; null: no flag array yet, Builder::materialise_flags)
                done[m + k] = 1;
            }
        }
    }
}

// flags + per-tile counts of what the finisher left: lf[m] = done ? 0 : active | (first of its group ? head : 0)
__global__ __launch_bounds__(BLD_BLOCK) void tiny_flags_kernel(const u32* __restrict__ gid, const u8* __restrict__ done, u32 n,
                                                               u8* __restrict__ lf, uint2* __restrict__ counts) {
    __shared__ u32 s_a[BLD_BLOCK / WAVE], s_h[BLD_BLOCK / WAVE];
    const u64 base = (u64)blockIdx.x * BLD_TILE;
    u32 ca = 0, ch = 0;
    for (int it = 0; it < BLD_ITEMS; ++it) {
        const u64 j = base + (u64)it * BLD_BLOCK + threadIdx.x;
        if (j < n) {
            const bool act = done[j] == 0;
            const bool head = (j == 0) || (gid[j - 1] != gid[j]);
            lf[j] = act ? (u8)(2 | (head ? 1 : 0)) : (u8)0;
            ca += act;
            ch += (act && head);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ca += __shfl_down(ca, o); ch += __shfl_down(ch, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_a[wave] = ca; s_h[wave] = ch; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 ta = 0, th = 0;
        for (int w = 0; w < BLD_BLOCK / WAVE; ++w) { ta += s_a[w]; th += s_h[w]; }
        counts[blockIdx.x] = make_uint2(ta, th);
    }
}

// ---- group starts (ranks) ----------------------------------------------------------------------------------
// last head value per tile (value = slot of the head: posmap[j] or j), NONE32 if the tile has none
__global__ __launch_bounds__(BLD_BLOCK) void tile_last_head_kernel(const u8* __restrict__ lf, u32 n,
                                                                   const u32* __restrict__ posmap,
                                                                   u32* __restrict__ tile_last) {
    __shared__ u32 s_w[BLD_BLOCK / WAVE];
    const u64 base = (u64)blockIdx.x * BLD_TILE;
    int64_t best = -1;  // largest j with a head
    for (int it = 0; it < BLD_ITEMS; ++it) {
        const u64 j = base + (u64)it * BLD_BLOCK + threadIdx.x;
        if (j < n && (lf[j] & 1u)) best = (int64_t)j;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t t = __shfl_down(best, o);
        best = t > best ? t : best;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int64_t s_b[BLD_BLOCK / WAVE];
    if (lane == 0) s_b[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t bb = -1;
        for (int w = 0; w < BLD_BLOCK / WAVE; ++w) bb = s_b[w] > bb ? s_b[w] : bb;
        tile_last[blockIdx.x] = (bb < 0) ? NONE32 : (posmap ? posmap[bb] : (u32)bb);
    }
    (void)s_w;
}

// exclusive "last defined value" scan over tiles, single workgroup:
// carry[t] = value of the last tile < t that has a head (NONE32 if none)
__global__ __launch_bounds__(1024) void scan_last_head_kernel(const u32* __restrict__ tile_last, u32 ntiles,
                                                              u32* __restrict__ carry) {
    __shared__ u32 s_v[1024];
    const u32 per = (ntiles + 1023) / 1024;
    const u32 lo = threadIdx.x * per;
    const u32 hi = (lo + per < ntiles) ? lo + per : ntiles;
    u32 v = NONE32;
    for (u32 i = lo; i < hi; ++i) { const u32 t = tile_last[i]; if (t != NONE32) v = t; }
    s_v[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        u32 t = NONE32;
        if ((int)threadIdx.x >= o) t = s_v[threadIdx.x - o];
        __syncthreads();
        if (s_v[threadIdx.x] == NONE32) s_v[threadIdx.x] = t;
        __syncthreads();
    }
    u32 c = (threadIdx.x == 0) ? NONE32 : s_v[threadIdx.x - 1];
    for (u32 i = lo; i < hi; ++i) {
        carry[i] = c;
        const u32 t = tile_last[i];
        if (t != NONE32) c = t;
    }
}

// isa[idx[j]] = slot of the head of j's group
__global__ __launch_bounds__(BLD_BLOCK) void scatter_ranks_kernel(const u8* __restrict__ lf, u32 n,
                                                                  const u32* __restrict__ posmap,
                                                                  const u32* __restrict__ idx,
                                                                  const u32* __restrict__ carry,
                                                                  u32* __restrict__ isa) {
    constexpr int WAVES = BLD_BLOCK / WAVE;
    __shared__ u32 s_last[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 le = lanemask_lt() | (1ull << lane);
    u32 run = carry[blockIdx.x];
    const u64 base = (u64)blockIdx.x * BLD_TILE;
    for (int it = 0; it < BLD_ITEMS; ++it) {
        const u64 j = base + (u64)it * BLD_BLOCK + threadIdx.x;
        const bool in = j < n;
        const bool head = in && (lf[j] & 1u);
        const u32 myval = in ? (posmap ? posmap[j] : (u32)j) : 0u;
        const u64 bh = __ballot(head);
        const u64 mine = bh & le;
        const int src = mine ? (63 - __clzll((long long)mine)) : 0;
        const u32 from_wave = __shfl(myval, src);
        const int lastlane = bh ? (63 - __clzll((long long)bh)) : 0;
        const u32 wave_last = __shfl(myval, lastlane);
        if (lane == 0) s_last[wave] = bh ? wave_last : NONE32;
        __syncthreads();
        u32 prev = run;  // last head before this wave in this iteration, else the running carry
        u32 newrun = run;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const u32 t = s_last[w];
            if (t != NONE32) {
                if (w < wave) prev = t;
                newrun = t;
            }
        }
        if (in) isa[idx[j]] = mine ? from_wave : prev;
        run = newrun;
        __syncthreads();
    }
}

// libsais64 layout (libsais64.c:6248-6259 widens in place on the CPU): out[i] = (int64)sa[i].  Four entries per
// thread and step: one 16-byte load, two 16-byte stores (device allocations are 16-byte aligned; an odd base
// address -- a caller's sub-range -- takes the element-wise loop).
__global__ __launch_bounds__(256) void widen_kernel(const u32* __restrict__ sa, u64 n, int64_t* __restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 t0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool aligned = ((reinterpret_cast<uintptr_t>(sa) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    u64 done = 0;
    if (aligned) {
        const u64 n4 = n / 4;
        const uint4* __restrict__ in4 = reinterpret_cast<const uint4*>(sa);
        uint4* __restrict__ out4 = reinterpret_cast<uint4*>(out);
        for (u64 i = t0; i < n4; i += stride) {
            const uint4 v = in4[i];
            out4[2 * i] = make_uint4(v.x, 0u, v.y, 0u);
            out4[2 * i + 1] = make_uint4(v.z, 0u, v.w, 0u);
        }
        done = n4 * 4;
    }
    for (u64 i = done + t0; i < n; i += stride) out[i] = (int64_t)sa[i];
}

// sa64[slot] = sa[slot] for the slots of a list (the suffixes that were still tied after the initial sort: whatever
// refined them wrote the u32 array only; the int64 copy that the sort's last pass left is brought up to date here)
__global__ void widen_patch_kernel(const u32* __restrict__ slots, u32 m, const u32* __restrict__ sa, int64_t* __restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const u32 s = slots[i];
        out[s] = (int64_t)sa[s];
    }
}

// entries of an adopted suffix array that cannot be suffix positions (>= n): sa_hip_index_load refuses the array
__global__ void sa_range_check_kernel(const u32* __restrict__ sa, u64 n, u64* __restrict__ bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 local = 0;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) local += (sa[j] >= n) ? 1u : 0u;
    if (local) atomicAdd((unsigned long long*)bad, (unsigned long long)local);
}

// ---- on-device verification (sufcheck) ----------------------------------------------
// SA is the suffix array of T  <=>  it is a permutation of [0,n) and for every adjacent pair
// (a, b) = (SA[j-1], SA[j]):  T[a] < T[b], or T[a] == T[b] and suffix a+1 sorts before suffix b+1
// (the empty suffix first) -- by induction over the suffix order the second test may use the
// ranks ISA[a+1] < ISA[b+1] of the array under test.  O(n) work, 4n bytes of scratch.
__global__ void verify_scatter_kernel(const u32* __restrict__ sa, u64 n, u32* __restrict__ isa, u64* __restrict__ bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const u32 v = sa[j];
        if (v >= n) atomicAdd((unsigned long long*)bad, 1ull);
        else isa[v] = (u32)j;
    }
}
__global__ void verify_order_kernel(const u8* __restrict__ text, const u32* __restrict__ sa, const u32* __restrict__ isa,
                                    u64 n, u64* __restrict__ bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 local = 0;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const u32 b = sa[j];
        if (b >= n) continue;                       // counted by the scatter kernel
        if (isa[b] != (u32)j) { ++local; continue; }   // duplicate value: not a permutation
        if (j == 0) continue;
        const u32 a = sa[j - 1];
        if (a >= n) continue;
        const u8 ca = text[a], cb = text[b];
        if (ca > cb) ++local;
        else if (ca == cb) {
            if ((u64)a + 1 == n) { /* empty suffix first: fine */ }
            else if ((u64)b + 1 == n) ++local;
            else if (isa[a + 1] >= isa[b + 1]) ++local;
        }
    }
    if (local) atomicAdd((unsigned long long*)bad, (unsigned long long)local);
}
// truncated order: adjacent suffixes compare <= on their first L bytes (a suffix that ends sorts
// first), ties in text order
__global__ void verify_truncated_kernel(const u8* __restrict__ text, const u32* __restrict__ sa, const u32* __restrict__ isa,
                                        u64 n, u32 L, u64* __restrict__ bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 local = 0;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const u32 b = sa[j];
        if (b >= n) continue;
        if (isa[b] != (u32)j) { ++local; continue; }
        if (j == 0) continue;
        const u32 a = sa[j - 1];
        if (a >= n) continue;
        int r = 0;
        for (u32 i = 0; i < L && r == 0; ++i) {
            const u64 pa = (u64)a + i, pb = (u64)b + i;
            const int xa = pa < n ? (int)text[pa] : -1, xb = pb < n ? (int)text[pb] : -1;
            if (xa != xb) r = xa < xb ? -1 : 1;
            else if (xa < 0) break;
        }
        if (r > 0 || (r == 0 && a > b)) ++local;
    }
    if (local) atomicAdd((unsigned long long*)bad, (unsigned long long)local);
}

// ---- debugging aid (SA_HIP_DEBUG_ROUNDS=1): duplicates in a list of suffix indices ------------------
// mark[v] = 1 + slot of the first occurrence; dups[0] = count, dups[1 + 2k], dups[2 + 2k] = the two slots
__global__ void dbg_dup_kernel(const u32* __restrict__ idx, u64 m, u32* __restrict__ mark, u64* __restrict__ dups) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const u32 old = atomicCAS(&mark[idx[j]], 0u, (u32)j + 1u);
        if (old != 0) {
            const u64 k = atomicAdd((unsigned long long*)dups, 1ull);
            if (k < 24) { dups[1 + 2 * k] = old - 1; dups[2 + 2 * k] = j; }
        }
    }
}

// ---- query acceleration structures (used by sa_query.hpp) -------------------------------------------
// keys[j] = packed first k0 characters of suffix sa[j] (for indexes adopted with sa_hip_index_load)
__global__ __launch_bounds__(256) void gather_keys_kernel(const u8* __restrict__ text, u64 n, CodeMap map, int b, int k0,
                                                          const u32* __restrict__ sa, u64* __restrict__ keys) {
    __shared__ u16 s_map[256];
    s_map[threadIdx.x] = map.code[threadIdx.x];
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const u64 start = sa[j];
        u64 key = 0;
        int sh = 64;
        for (int i = 0; i < k0; ++i) {
            const u64 p = start + i;
            sh -= b;
            const u64 c = (p < n) ? (u64)s_map[text[p]] : 0ull;
            key |= c << sh;
        }
        keys[j] = key;
    }
}

// dir[bkt] = first slot whose key has top-dbits >= bkt, by binary search; dir[2^dbits] = n.
// `coarse` (may be null) is a directory over the top cbits < dbits: the search then starts inside the
// coarse bucket that contains the boundary instead of [0, n).
__global__ __launch_bounds__(256) void dir_build_kernel(const u64* __restrict__ keys, u64 n, int dbits, u32* __restrict__ dir,
                                                        const u32* __restrict__ coarse, int cbits) {
    const u64 nb = (1ull << dbits) + 1;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 bkt = (u64)blockIdx.x * blockDim.x + threadIdx.x; bkt < nb; bkt += stride) {
        u64 lo = 0, hi = n;
        if (bkt == nb - 1) lo = n;
        else {
            const u64 bound = bkt << (64 - dbits);
            if (coarse) {
                const u64 cb = bkt >> (dbits - cbits);
                lo = coarse[cb];
                hi = coarse[cb + 1];
            }
            while (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if (keys[mid] < bound) lo = mid + 1; else hi = mid;
            }
        }
        dir[bkt] = (u32)lo;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        SA_HIP_CHECK(hipMalloc(&p, bytes));
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline u32 stream_grid(u64 work_items, u32 per_block) {
    u64 g = (work_items + per_block - 1) / per_block;
    if (g > 256u * 8u) g = 256u * 8u;
    if (g == 0) g = 1;
    return (u32)g;
}

#ifndef SA_FLAGS_GRID
#define SA_FLAGS_GRID 32768
#endif

struct Builder {
    hipStream_t stream = nullptr;
    u64 n_max = 0;
    // persistent buffers
    DevBuf text, keys0, keys1, vals0, vals1, flags, counts, small, isa;
    // round pool
    DevBuf apos0, apos1, apos2, aidx, gid, rkeys0, rkeys1, ridx0, ridx1, lf, tile_last, carry;
    RadixWorkspace radix;
    NarrowWorkspace narrow;
    bool partial_char = true;         // SA_HIP_PARTIAL_CHAR: the 10-byte plan fills its 56 key bits with the top bits of one more character
    bool narrow48 = true;             // SA_HIP_NARROW48: 10-byte records for initial keys of 41..56 bits (radix_narrow48.hpp)
    u32* sa = nullptr;        // points into vals0/vals1 after a build (or into sa_own after load)
    DevBuf sa_own;
    u64 n = 0;
    u32 max_suffix_length = 0;
    u64 freq[256];
    sa_hip_build_stats stats;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    int chunk_rounds_before_doubling = 2;
    bool adaptive_doubling = true;    // SA_HIP_ADAPTIVE_DOUBLING: more chunk rounds while the active set halves
    int initial_chars_override = 0;   // SA_HIP_INITIAL_CHARS: 0 = heuristic
    double pilot_dup_share = 0.0;     // share of sampled 8-byte windows seen before (pilot_kernel)
    DevBuf pilot;
    bool fuse_hist = true;            // SA_HIP_FUSE_HIST: digit histograms inside keygen
    bool fuse_directory = true;       // SA_HIP_FUSE_DIR: query directory written by the first flags pass
    bool text_top_pass = true;        // SA_HIP_TEXT_PASS: the top-digit pass of a narrow sort builds its keys from the text
    bool wide_text_pass = true;       // SA_HIP_WIDE_TEXT_PASS: pass 0 of the 12-byte-record sort builds its keys from the text
    bool narrow_sort = true;          // SA_HIP_NARROW: 8-byte records for initial keys of <= 40 bits (radix_narrow.hpp)
    DevBuf partial, dbg, done;
    bool tiny_finisher = true;        // SA_HIP_TINY: direct-comparison finisher for groups of <= 8
    bool local_rounds = true;         // SA_HIP_LOCAL_ROUNDS: rounds sorted group-wise in LDS (round_sort.hpp)
    DevBuf gstart, loc_tiles, big_keys, big_vals;
    bool group_finish = true;         // SA_HIP_GROUP_FINISH: groups that fit a tile are refined to the end in LDS (group_finish.hpp)
    bool fin_left_failed = false;     // this build: a finisher run has given up on a group (the shortcut below is not tried again)
    bool fin_left_fast = true;        // SA_HIP_FIN_LEFT_FAST=0: what a finisher run leaves is always found from the done flags (flags pass + scan + compaction over the list)
    DevBuf fin_left_gid;              // u32[tiles]: next dense group id of a tile's left-out group
    bool fin_prefetch = false;        // SA_HIP_FIN_PREFETCH=1: the finisher's first-round text fetches as a kernel of their own (measured slower: the gather costs what it saves)
    bool fin_v2 = false;              // SA_HIP_FIN_V2=1: round 4's restructured finisher (group_finish2_kernel); measured 0-2 % SLOWER than the round-2 kernel
                                      // (profiles/r04_finisher_v2_ab.log), so it is not the default
    bool fin_useful = true;           // per build: cleared when a run resolves less than a quarter of what it looked at
    bool use_pilot = true;            // SA_HIP_PILOT: 0 = initial key length from the byte distribution alone
    DevBuf fin_flag;                  // u8[M]: per list position, final / head marks of the finisher
    DevBuf fin_w0;                    // u64[M]: per list position, the text bytes of the finisher's first round (fin_prefetch_kernel)
    bool period_finish = true;        // SA_HIP_PERIOD_FINISH: arithmetic groups inside one periodic run are ordered in one step (period_finish.hpp)
    int per_skip = 0, per_fails = 0;  // per build: an attempt that orders less than an eighth of the active set is repeated only after 2, 4, 8 ... rounds
    DevBuf per_gd, per_bad, per_table, per_dec, per_tf, per_carry;
    // the slot lists of the active set: the next compaction writes into lst_nxt.  With an int64 copy to keep up to date the
    // FIRST list (apos0) is left alone -- it names every slot that is written after the sort -- and later lists
    // ping-pong between apos1 and apos2.
    u32* lst_cur = nullptr;
    u32* lst_nxt = nullptr;
    bool lst_first = true, lst_keep_first = false;
    void swap_lists() {
        if (lst_keep_first && lst_first) { lst_cur = lst_nxt; lst_nxt = apos2.as<u32>(); }
        else { u32* t = lst_cur; lst_cur = lst_nxt; lst_nxt = t; }
        lst_first = false;
    }
    int big_round_chars = 0;             // SA_HIP_BIG_ROUND_CHARS: characters per global round while the finisher is at work (0 = as many as fit; 3..7 measured: no gain, tools/gpu_bigchars_sweep.py)
    u32 fin_count_max = FIN_COUNT_MAX;   // SA_HIP_FIN_COUNT_MAX
    int fin_radix_chars = FIN_RADIX_CHARS;   // SA_HIP_FIN_RADIX_CHARS
    u64 local_records = 0, big_records = 0;   // of the last build: records sorted in LDS / through the big-group list
    LocTile one_tile{};                       // source of an asynchronous copy (round_sort, lists of at most one tile)
    bool debug_rounds = false;
    // query acceleration (sa_query.hpp): sorted packed keys K + bucket directory
    const u64* qkeys = nullptr;   // points into keys0/keys1 (build) or keys0 (load)
    // Second-level keys (round 4, sa_query.hpp: k2_build_kernel): for every SA slot whose key equals a neighbour's, the q_k2n
    // characters that FOLLOW the key's q_k0, packed like the key.  Built on demand (the first large batch of a wide-key index,
    // or sa_hip_index_deep_keys), 8 n bytes; k2_ready falls with every change of qkeys.
    DevBuf qkeys2;
    DevBuf qskeys;   // every 64th key of qkeys (sa_query.hpp: SKEY_STRIDE), built with qkeys2
    bool k2_ready = false;
    int q_k2n = 0;
    // ... or, after a narrow-record sort (narrow_k): u32 narrow keys + the 257 bucket bounds of the top digit;
    // key of slot j = (bucket(j) << 56) | (qkeys32[j] << q_lo_shift).  Exactly one of qkeys / qkeys32 is set.
    const u32* qkeys32 = nullptr;
    const u32* q_bstart = nullptr;
    int q_lo_shift = 0;
    bool narrow_k = true;         // SA_HIP_NARROW_K: keep the narrow keys (the last narrow pass writes 4-byte keys)
    DevBuf qdir;
    CodeMap qmap;
    int q_b = 0, q_k0 = 0, q_dbits = 0;
    bool dir_ready = false;   // the directory in qdir belongs to qkeys
    bool dir_by_local_pass = false;   // ... and was written by the three-pass plan's local pass, at the end of the sort
    int sector_search = 2;        // SA_HIP_SECTOR_SEARCH: interpolated scan inside a directory bucket (sa_query.hpp): 1 = 64-byte windows, 2 = 32-byte windows, 0 = binary search

    // Characters in the initial key.  Enough that, for an i.i.d. text with this byte
    // distribution, about 2 % of the suffixes still share their key (collision probability
    // per character c2 = sum p_i^2), then rounded up to fill whole 8-bit sort passes.
    // the 10-byte-record plan can be taken at all (the key length decides later whether it is)
    bool narrow48_possible(int b) const {
        return narrow_sort && narrow48 && fuse_hist && text_top_pass && radix.block == 512 && b <= 8 && n >= (1u << 22) &&
               n / (512u * SEG48_LAST_ITEMS) + RADIX + 1 <= radix.max_tiles;
    }
    int choose_initial_chars(int b, u32 L) const {
        // the longest key: 64 bits as 12-byte records, or -- where the 10-byte-record plan applies -- 56 bits (for word / name
        // texts one character less in the key costs less than 4 bytes per record and pass: config 5 70.1 vs 70.2 ms at
        // 11 / 12 characters on 12-byte records)
        const int kmax = (narrow48_possible(b) && 56 / b >= 6) ? 56 / b : 64 / b;
        int k = kmax;
        if (initial_chars_override > 0) {
            k = initial_chars_override < kmax ? initial_chars_override : kmax;
        } else {
            double c2 = 0.0;
            for (int c = 0; c < 256; ++c) { const double pr = (double)freq[c] / (double)n; c2 += pr * pr; }
            if (c2 < 0.999999) {
                const double need = std::log(0.02 / (double)n) / std::log(c2);
                int kk = (int)std::ceil(need);
                if (kk < 1) kk = 1;
                // the pilot's 8-byte windows repeat in an i.i.d. text too when the alphabet is small (4 symbols:
                // 65 536 possible windows for 131 072 samples): compare with the share such a text would show
                const double d_eff = 1.0 / std::pow(c2, 8.0);   // effective number of distinct windows
                const double s_n = (double)PILOT_SAMPLES;
                const double iid_share = 1.0 + d_eff * std::expm1(-s_n / d_eff) / s_n;
                if (kk < kmax && pilot_dup_share <= iid_share + 0.03) {
                    const int passes = (kk * b + RADIX_BITS - 1) / RADIX_BITS;
                    kk = (passes * RADIX_BITS) / b;
                    k = kk < kmax ? kk : kmax;
                }
            }
        }
        if (L && (u32)k > L) k = (int)L;
        return k < 1 ? 1 : k;
    }

    // Small device-to-host reads go through a pinned mailbox: a hipMemcpyAsync into PAGEABLE memory blocks the host until the copy
    // has landed, so reading two values cost two round trips (the kernel trace of a build shows ~20 us of idle GPU per such copy).
    u8* mbox = nullptr;
    enum : size_t { MB_FREQ = 0, MB_PILOT = 2048, MB_TOTALS = 2064, MB_STATUS = 2112, MB_BIG = 2240, MB_FT = 2304, MB_BEST = 2560, MB_BYTES = 4096 };
    int init(u64 nmax, hipStream_t s) {
        stream = s;
        n_max = nmax;
        if (!mbox) SA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&mbox), MB_BYTES, hipHostMallocDefault));
        const u64 cap = nmax ? nmax : 1;
        int rc;
        if ((rc = text.ensure(cap + TEXT_PAD + 16))) return rc;
        if ((rc = small.ensure(4096))) return rc;
        int sort_block = 512;
        if (const char* e = diag_env("SA_HIP_SORT_BLOCK")) sort_block = (atoi(e) == 256) ? 256 : 512;
        if (const char* e = diag_env("SA_HIP_INITIAL_CHARS")) initial_chars_override = atoi(e);
        if (const char* e = diag_env("SA_HIP_CHUNK_ROUNDS")) chunk_rounds_before_doubling = atoi(e);
        if (const char* e = diag_env("SA_HIP_ADAPTIVE_DOUBLING")) adaptive_doubling = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_FUSE_HIST")) fuse_hist = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_FUSE_DIR")) fuse_directory = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_NARROW")) narrow_sort = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_NARROW48")) narrow48 = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_PARTIAL_CHAR")) partial_char = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_TEXT_PASS")) text_top_pass = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_WIDE_TEXT_PASS")) wide_text_pass = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_NARROW_K")) narrow_k = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_TINY")) tiny_finisher = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_LITE_FLAGS")) lite_flags = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_LOCAL_ROUNDS")) local_rounds = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_GROUP_FINISH")) group_finish = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_FIN_V2")) fin_v2 = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_FIN_LEFT_FAST")) fin_left_fast = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_FIN_PREFETCH")) fin_prefetch = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_PILOT")) use_pilot = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_PERIOD_FINISH")) period_finish = atoi(e) != 0;
        if (const char* e = diag_env("SA_HIP_SECTOR_SEARCH")) sector_search = atoi(e);
        if (const char* e = diag_env("SA_HIP_FIN_COUNT_MAX")) fin_count_max = (u32)atoi(e);
        if (const char* e = diag_env("SA_HIP_BIG_ROUND_CHARS")) big_round_chars = atoi(e);
        if (const char* e = diag_env("SA_HIP_FIN_RADIX_CHARS")) fin_radix_chars = atoi(e);
        if (const char* e = diag_env("SA_HIP_DEBUG_ROUNDS")) debug_rounds = atoi(e) != 0;
        if (debug_rounds) { radix.debug_hook = &Builder::sort_debug_hook; radix.debug_ctx = this; }
        if ((rc = radix.init(cap, sort_block))) return rc;
        if ((rc = narrow.init())) return rc;
        SA_HIP_CHECK(hipEventCreate(&ev_begin));
        SA_HIP_CHECK(hipEventCreate(&ev_end));
        memset(&stats, 0, sizeof stats);
        memset(freq, 0, sizeof freq);
        return 0;
    }
    // the ping-pong key buffers: u64[n] for the 12-byte-record plan, u32[n] when the sort runs on narrow records (8 GB less
    // at n = 1e9); + one sector: the query path reads whole 64-byte sectors of the key array
    int ensure_key_buffers(u64 count, bool narrow) {
        const size_t bytes = (size_t)(count ? count : 1) * (narrow ? 4 : 8) + 64;
        int rc;
        if ((rc = keys0.ensure(bytes))) return rc;
        return keys1.ensure(bytes);
    }
    int ensure_build_buffers() {
        const u64 cap = n_max ? n_max : 1;
        int rc;
        if ((rc = vals0.ensure(cap * 4))) return rc;
        if ((rc = vals1.ensure(cap * 4))) return rc;
        if ((rc = flags.ensure(cap + 64))) return rc;
        if ((rc = counts.ensure((size_t)div_up(cap, BLD_TILE) * sizeof(uint2) + 64))) return rc;
        return 0;
    }
    void destroy() {
        DevBuf* all[] = {&text, &keys0, &keys1, &vals0, &vals1, &flags, &counts, &small, &isa, &apos0, &apos1, &apos2, &aidx,
                         &gid, &rkeys0, &rkeys1, &ridx0, &ridx1, &lf, &tile_last, &carry, &sa_own, &partial, &qdir, &dbg, &done, &pilot,
                         &gstart, &loc_tiles, &big_keys, &big_vals, &fin_flag, &fin_w0, &fin_left_gid, &lite_stage, &qkeys2, &qskeys, &per_gd, &per_bad, &per_table, &per_dec, &per_tf, &per_carry};
        for (DevBuf* b : all) b->release();
        radix.destroy();
        narrow.destroy();
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
        ev_begin = ev_end = nullptr;
        if (mbox) (void)hipHostFree(mbox);
        mbox = nullptr;
    }

    // text must already be in text.p[0..n); pads it and computes freq + code map
    int prepare_text(u64 n_, CodeMap& map, u32& sigma, int& b) {
        n = n_;
        SA_HIP_CHECK(hipMemsetAsync(text.as<u8>() + n, 0, TEXT_PAD + 16, stream));
        u64* dh = small.as<u64>();
        SA_HIP_CHECK(hipMemsetAsync(dh, 0, 256 * sizeof(u64), stream));
        if (n) hipLaunchKernelGGL(byte_hist_kernel, dim3(stream_grid(n, 256 * 64)), dim3(256), 0, stream, text.as<u8>(), n, dh);
        static_assert(sizeof freq == 2048, "mailbox layout");
        SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_FREQ, dh, sizeof freq, hipMemcpyDeviceToHost, stream));
        // pilot (independent of the histogram: raw bytes), only for texts large enough to matter
        u32 pilot_dups = 0;
        pilot_dup_share = 0.0;
        const bool run_pilot = use_pilot && n >= (u64)16 * PILOT_SAMPLES;
        if (run_pilot) {
            int rc = pilot.ensure((size_t)PILOT_SLOTS * 8 + 64);
            if (rc) return rc;
            SA_HIP_CHECK(hipMemsetAsync(pilot.p, 0, (size_t)PILOT_SLOTS * 8 + 64, stream));
            u32* pd = reinterpret_cast<u32*>(pilot.as<u8>() + (size_t)PILOT_SLOTS * 8);
            hipLaunchKernelGGL(pilot_kernel, dim3(PILOT_SAMPLES / 256), dim3(256), 0, stream, text.as<u8>(), n, 8,
                               pilot.as<unsigned long long>(), pd);
            SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_PILOT, pd, 4, hipMemcpyDeviceToHost, stream));
        }
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        memcpy(freq, mbox + MB_FREQ, sizeof freq);
        if (run_pilot) memcpy(&pilot_dups, mbox + MB_PILOT, 4);
        if (run_pilot) pilot_dup_share = (double)pilot_dups / (double)PILOT_SAMPLES;
        sigma = 0;
        memset(&map, 0, sizeof map);
        for (int c = 0; c < 256; ++c)
            if (freq[c]) map.code[c] = (u16)(++sigma);
        b = bits_for((u64)sigma + 1);
        if (b == 0) b = 1;
        return 0;
    }

    // the totals and, in the same synchronisation, the sort's device status (check_device_status() looks at this copy:
    // one host round trip per flags pass instead of two)
    DeviceStatus status_seen{};
    u32 lite_overflow_seen = 0;
    bool status_fresh = false;
    int read_totals(u32* totals_host) {
        static_assert(sizeof(DeviceStatus) <= MB_BIG - MB_STATUS, "mailbox layout");
        SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_TOTALS, small.as<u8>() + 2048, 3 * sizeof(u32), hipMemcpyDeviceToHost, stream));   // {active, heads, lite overflow}
        SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_STATUS, radix.dstat, sizeof status_seen, hipMemcpyDeviceToHost, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        memcpy(totals_host, mbox + MB_TOTALS, 2 * sizeof(u32));
        memcpy(&lite_overflow_seen, mbox + MB_TOTALS + 2 * sizeof(u32), sizeof(u32));
        memcpy(&status_seen, mbox + MB_STATUS, sizeof status_seen);
        status_fresh = true;
        return 0;
    }
    u32* totals_dev() { return reinterpret_cast<u32*>(small.as<u8>() + 2048); }

    int check_device_status() {
        DeviceStatus st;
        if (status_fresh) {   // read together with the totals of the pass that has just been synchronised
            st = status_seen;
            status_fresh = false;
        } else {
            SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_STATUS, radix.dstat, sizeof st, hipMemcpyDeviceToHost, stream));
            SA_HIP_CHECK(hipStreamSynchronize(stream));
            memcpy(&st, mbox + MB_STATUS, sizeof st);
        }
        if (st.error != 0) {
            (void)hipMemsetAsync(radix.dstat, 0, sizeof(DeviceStatus), stream);
            return fail(SA_HIP_EINTERNAL, "device look-back spin limit expired");
        }
        return 0;
    }

    // exclusive scan of counts[0..tiles) + totals to the host
    int scan_counts(u32 tiles, u32* totals_host) {
        const u32 parts = div_up(tiles, 1024);
        int rc = partial.ensure((size_t)parts * sizeof(uint2) + 64);
        if (rc) return rc;
        hipLaunchKernelGGL(counts_reduce_kernel, dim3(parts), dim3(1024), 0, stream, counts.as<uint2>(), tiles, partial.as<uint2>());
        hipLaunchKernelGGL(counts_scan_partials_kernel, dim3(1), dim3(1024), 0, stream, partial.as<uint2>(), parts, totals_dev());
        hipLaunchKernelGGL(counts_apply_kernel, dim3(parts), dim3(1024), 0, stream, counts.as<uint2>(), tiles, partial.as<uint2>());
        return read_totals(totals_host);
    }

    // compaction of the active elements of a domain of `cnt` elements, `active` of them active
    void launch_compact(const u8* lf_in, u32 cnt, u32 active, const u32* src_pos, const u32* src_idx, u32* dst_pos,
                        u32* dst_idx, u32* dst_gid) {
        const u32 tiles = div_up(cnt, BLD_TILE);
#ifndef SA_DENSE_DIV
#define SA_DENSE_DIV 100
#endif
        // >= 1 % active: the lane-strided variant.  The sparse one costs ~0.5 ps per slot + ~16 ps per active element (D1 at n = 1e9,
        // 0.35 % active: 0.565 ms against 0.670 ms for the dense one; 10 % active after a finisher run: 2.1 ps per slot against ~0.8),
        // the crossover lies near 1 % (12.5 % until the dense variant lost its per-row barriers)
        if ((u64)active * SA_DENSE_DIV >= cnt)
            hipLaunchKernelGGL(compact_dense_kernel, dim3(tiles), dim3(BLD_BLOCK), 0, stream, lf_in, cnt, counts.as<uint2>(), src_pos,
                               src_idx, dst_pos, dst_idx, dst_gid);
        else
            hipLaunchKernelGGL(compact_kernel, dim3(tiles), dim3(BLD_BLOCK), 0, stream, lf_in, cnt, counts.as<uint2>(), src_pos,
                               src_idx, dst_pos, dst_idx, dst_gid);
    }

    // head/active flags over `cnt` sorted keys (+ optional round write-back), scanned counts.
    int flags_and_counts(const u64* keys, u32 cnt, u8* lf_out, const u32* apos, const u32* sidx, u32* totals_host,
                         bool with_directory = false, const NarrowKeys* nk = nullptr) {
        const u32 tiles = div_up(cnt, BLD_TILE);
        DirArgs d{};
        if (with_directory) {
            // the directory of the query path comes out of this pass over the sorted keys for free
            int rc = directory_layout(cnt);
            if (rc) return rc;
            d.dir = qdir.as<u32>();
            d.dbits = q_dbits;
            const u64 nb = (1ull << q_dbits) + 1;
            d.gaps = reinterpret_cast<uint4*>(qdir.as<u8>() + dir_gap_offset(nb));
            d.gap_cap = dir_gap_cap(nb);
            d.gap_count = reinterpret_cast<u32*>(small.as<u8>() + 3584);
            d.dstat = radix.dstat;
            SA_HIP_CHECK(hipMemsetAsync(d.gap_count, 0, 4, stream));
        }
        // a resident-sized grid that walks the tiles instead of one short-lived workgroup per 4096 slots (244 141 of them at n = 1e9):
        // flags_kernel<true> 2.0 -> 1.55 ms at n = 1e9 with 32768 workgroups (4096 / 8192 / 16384 / 65536: 1.78 / 1.8 / 1.88 / 1.72;
        // 2048, the exact resident count: 2.24), -6 % at 5e8, -18 % at 7.77e8, -9 % at 2e9 (profiles/r03_flags_grid.log)
        const u32 fgrid = (SA_FLAGS_GRID && tiles > (u32)SA_FLAGS_GRID) ? (u32)SA_FLAGS_GRID : tiles;
        if (nk)
            hipLaunchKernelGGL(flags_kernel<true>, dim3(fgrid), dim3(BLD_BLOCK), 0, stream, (const u64*)nullptr, *nk, cnt, lf_out,
                               counts.as<uint2>(), apos, sidx, sa, flags.as<u8>(), d);
        else
            hipLaunchKernelGGL(flags_kernel<false>, dim3(fgrid), dim3(BLD_BLOCK), 0, stream, keys, NarrowKeys{}, cnt, lf_out,
                               counts.as<uint2>(), apos, sidx, sa, flags.as<u8>(), d);
        if (with_directory) {
            hipLaunchKernelGGL(dir_fill_kernel, dim3(1024), dim3(256), 0, stream, d);
            dir_ready = true;
        }
        return scan_counts(tiles, totals_host);
    }

    // The first flags pass in its lite form (flags_lite_kernel): directory + per-tile counts + the active records staged in `stage`
    // (a buffer of at least ntiles * LITE_CAP * 9 bytes that is free after the sort).  No flag array: flags_valid stays false.
    bool lite_flags = true;           // SA_HIP_LITE_FLAGS
    bool flags_valid = true;          // the flag array holds the head bits of the current grouping
    LiteArgs lite{};
    NarrowKeys lite_nk{};
    // The same work done by the local pass of the three-pass plan (radix_split.hpp: local_finish_kernel<., true>) on the
    // sub-bucket it holds in LDS: this is what the sort calls once it knows how many sub-buckets there are.  The staging rows
    // cannot live in a key buffer here (the pass still reads one and writes the other): a buffer of their own.
    DevBuf lite_stage;
    DirArgs split_dir{};
    u32 lite_tiles = 0;   // rows of the staging area: tiles of the lite pass, or sub-buckets
    static int split_flags_prepare(void* ctx, u32 nsub, LocalArgs* l) {
        Builder* b = static_cast<Builder*>(ctx);
        int rc = b->directory_layout(b->n);
        if (rc) return rc;
        if ((rc = b->counts.ensure((size_t)nsub * sizeof(uint2) + 64))) return rc;
        if ((rc = b->lite_stage.ensure((size_t)nsub * LITE_CAP * 9 + 64))) return rc;
        DirArgs d{};
        d.dir = b->qdir.as<u32>();
        d.dbits = b->q_dbits;
        const u64 nb = (1ull << b->q_dbits) + 1;
        d.gaps = reinterpret_cast<uint4*>(b->qdir.as<u8>() + dir_gap_offset(nb));
        d.gap_cap = dir_gap_cap(nb);
        d.gap_count = reinterpret_cast<u32*>(b->small.as<u8>() + 3584);
        d.dstat = b->radix.dstat;
        SA_HIP_CHECK(hipMemsetAsync(d.gap_count, 0, 4, b->stream));
        b->lite.sa = nullptr;
        b->lite.st_pos = b->lite_stage.as<u32>();
        b->lite.st_idx = b->lite.st_pos + (size_t)nsub * LITE_CAP;
        b->lite.st_head = reinterpret_cast<u8*>(b->lite.st_idx + (size_t)nsub * LITE_CAP);
        b->lite.overflow = b->totals_dev() + 2;
        SA_HIP_CHECK(hipMemsetAsync(b->lite.overflow, 0, 4, b->stream));
        SA_HIP_CHECK(hipMemsetAsync(b->counts.p, 0, (size_t)nsub * sizeof(uint2), b->stream));   // the local pass adds to the pairs of the sub-buckets that hold ties
        b->split_dir = d;
        b->lite_tiles = nsub;
        l->dir = d;
        l->counts = b->counts.as<uint2>();
        l->lite = b->lite;
        return 0;
    }
    // ... and what is left to do after such a sort: the scan of the per-sub-bucket counts
    int flags_after_split(const NarrowKeys& nk, u32* totals_host, bool* overflow) {
        dir_ready = true;   // (every directory entry was written by the local pass: no queued runs)
        dir_by_local_pass = true;
        lite_nk = nk;
        int rc = scan_counts(lite_tiles, totals_host);
        if (rc) return rc;
        *overflow = lite_overflow_seen != 0;
        return 0;
    }
    int flags_lite_pass(const NarrowKeys& nk, u32 cnt, void* stage, u32* totals_host, bool* overflow) {
        const u32 tiles = div_up(cnt, BLD_TILE);
        lite_tiles = tiles;
        int rc = directory_layout(cnt);
        if (rc) return rc;
        DirArgs d{};
        d.dir = qdir.as<u32>();
        d.dbits = q_dbits;
        const u64 nb = (1ull << q_dbits) + 1;
        d.gaps = reinterpret_cast<uint4*>(qdir.as<u8>() + dir_gap_offset(nb));
        d.gap_cap = dir_gap_cap(nb);
        d.gap_count = reinterpret_cast<u32*>(small.as<u8>() + 3584);
        d.dstat = radix.dstat;
        SA_HIP_CHECK(hipMemsetAsync(d.gap_count, 0, 4, stream));
        lite.sa = sa;
        lite.st_pos = static_cast<u32*>(stage);
        lite.st_idx = lite.st_pos + (size_t)tiles * LITE_CAP;
        lite.st_head = reinterpret_cast<u8*>(lite.st_idx + (size_t)tiles * LITE_CAP);
        lite.overflow = totals_dev() + 2;
        lite_nk = nk;
        SA_HIP_CHECK(hipMemsetAsync(lite.overflow, 0, 4, stream));
        SA_HIP_CHECK(hipMemsetAsync(counts.p, 0, (size_t)tiles * sizeof(uint2), stream));   // the threads add their {active, heads} to their tile's pair
        const u32 fgrid = (SA_FLAGS_GRID && tiles > (u32)SA_FLAGS_GRID) ? (u32)SA_FLAGS_GRID : tiles;
        hipLaunchKernelGGL(flags_lite_kernel, dim3(fgrid), dim3(BLD_BLOCK), 0, stream, nk, cnt, counts.as<uint2>(), d, lite);
        hipLaunchKernelGGL(dir_fill_kernel, dim3(1024), dim3(256), 0, stream, d);
        dir_ready = true;
        if ((rc = scan_counts(tiles, totals_host))) return rc;
        *overflow = lite_overflow_seen != 0;
        return 0;
    }
    // the flag array after a lite pass, for the phases that mark heads in it: head bits of the initial grouping from the kept key
    // array (one more pass over it; the counts it also produces are not used)
    int materialise_flags() {
        if (flags_valid) return 0;
        const u32 n32 = (u32)n;
        const u32 tiles = div_up(n32, BLD_TILE);
        const u32 fgrid = (SA_FLAGS_GRID && tiles > (u32)SA_FLAGS_GRID) ? (u32)SA_FLAGS_GRID : tiles;
        hipLaunchKernelGGL(flags_kernel<true>, dim3(fgrid), dim3(BLD_BLOCK), 0, stream, (const u64*)nullptr, lite_nk, n32, flags.as<u8>(),
                           counts.as<uint2>(), (const u32*)nullptr, (const u32*)nullptr, sa, flags.as<u8>(), DirArgs{});
        flags_valid = true;
        SA_HIP_CHECK(hipGetLastError());
        return 0;
    }

    // isa[idx[j]] = head slot of j's group over a domain of cnt elements
    int scatter_ranks(const u8* lf_in, u32 cnt, const u32* posmap, const u32* idx) {
        const u32 tiles = div_up(cnt, BLD_TILE);
        int rc;
        if ((rc = tile_last.ensure((size_t)tiles * 4 + 64))) return rc;
        if ((rc = carry.ensure((size_t)tiles * 4 + 64))) return rc;
        hipLaunchKernelGGL(tile_last_head_kernel, dim3(tiles), dim3(BLD_BLOCK), 0, stream, lf_in, cnt, posmap, tile_last.as<u32>());
        hipLaunchKernelGGL(scan_last_head_kernel, dim3(1), dim3(1024), 0, stream, tile_last.as<u32>(), tiles, carry.as<u32>());
        hipLaunchKernelGGL(scatter_ranks_kernel, dim3(tiles), dim3(BLD_BLOCK), 0, stream, lf_in, cnt, posmap, idx,
                           carry.as<u32>(), isa.as<u32>());
        return 0;
    }

    static void sort_debug_hook(void* ctx, int pass, int npasses, int shift, u32 mask, const u64* kin, const u32* vin,
                                const u64* keys, const u32* vals, u32 cnt) {
        Builder* b = static_cast<Builder*>(ctx);
        if (b->dbg.ensure((size_t)b->n * 4 + 16)) return;
        u64* dd = reinterpret_cast<u64*>(b->small.as<u8>() + 3072);
        u64 dups[49];
        (void)hipMemsetAsync(b->dbg.p, 0, (size_t)b->n * 4, b->stream);
        (void)hipMemsetAsync(dd, 0, sizeof dups, b->stream);
        hipLaunchKernelGGL(dbg_dup_kernel, dim3(stream_grid(cnt, 1024)), dim3(256), 0, b->stream, vals, (u64)cnt, b->dbg.as<u32>(), dd);
        (void)hipMemcpyAsync(dups, dd, sizeof dups, hipMemcpyDeviceToHost, b->stream);
        (void)hipStreamSynchronize(b->stream);
        if (dups[0]) {
            fprintf(stderr, "[sa_hip]   sort pass %d/%d over %u records: %llu duplicate values; slot pairs:", pass, npasses, cnt, (unsigned long long)dups[0]);
            for (u64 k = 0; k < dups[0] && k < 24; ++k) fprintf(stderr, " (%llu,%llu)", (unsigned long long)dups[1 + 2 * k], (unsigned long long)dups[2 + 2 * k]);
            fprintf(stderr, "\n");
        }
        (void)keys;
    }

    // Bucket directory over the top q_dbits bits of the sorted keys (qkeys must be set), built in two
    // levels: a 2^14-bucket directory by full binary searches, then the fine one inside its buckets.
    // Default q_dbits = log2(n) - 3, at most 27 (round 3: about 8 slots per bucket = one 32-byte window of narrow keys, the
    // granule random reads are served in, sa_query.hpp; at n = 1e9 with 32-byte windows, per 1M-query batch, 25 / 26 / 27 / 28
    // bits: 0.108 / 0.090 / 0.082 / 0.083 ms, build 21.0 / 21.26 / 21.5 / 27.1 ms on the same box -- the directory entries
    // are written by the flags pass, 4 bytes each, 537 MB at 27 bits; at 28 bits the 1 GB of scattered directory stores cost
    // more than the batch gains); SA_HIP_DIR_BITS overrides.
    // qdir layout: fine directory u32[nb] | coarse directory u32[ncb] (two-level build only) | gap queue
    static int dir_coarse_bits(int d) { return d > 16 ? 14 : 0; }
    static size_t dir_gap_offset(u64 nb) { return (((size_t)nb + (1u << 14) + 1) * 4 + 15) & ~(size_t)15; }
    // a queued run holds more than DIR_INLINE buckets and runs are disjoint, so nb / DIR_INLINE entries suffice
    // (+ one more entry per DIR_PIECE buckets for the pieces of long runs)
    static u32 dir_gap_cap(u64 nb) { return (u32)(nb / DIR_INLINE + nb / DIR_PIECE + 4); }
    int directory_layout(u64 count) {
        int lg = 0;
        while ((1ull << lg) < count) ++lg;
        int d = lg - 3;
        if (d < 8) d = 8;
        if (d > 27) d = 27;
        if (const char* e = diag_env("SA_HIP_DIR_BITS")) { const int v = atoi(e); if (v >= 8 && v <= 28) d = v; }
        q_dbits = d;
        const u64 nb = (1ull << d) + 1;
        return qdir.ensure(dir_gap_offset(nb) + (size_t)dir_gap_cap(nb) * sizeof(uint4));
    }

    // Directory by binary search (adopted indexes; a build gets it from its first flags pass)
    int build_directory() {
        if ((!qkeys && !qkeys32) || n < 2) { qkeys = nullptr; qkeys32 = nullptr; k2_ready = false; return 0; }
        if (qkeys32 && !dir_ready) return fail(SA_HIP_EINTERNAL, "narrow keys without a fused directory");
        if (dir_ready) {
            // (also worth it beyond the cache's 256 MiB: at 26 bits the first batch after a build takes 0.125 ms with
            //  this read, 0.147 ms without, and the build time is the same within the noise)
            const u64 n16 = ((1ull << q_dbits) + 1) / 4;
            if (const char* e = diag_env("SA_HIP_DIR_TOUCH")) { if (atoi(e) == 0) return 0; }
            hipLaunchKernelGGL(dir_touch_kernel, dim3(stream_grid(n16, 256)), dim3(256), 0, stream, qdir.as<uint4>(), n16,
                               reinterpret_cast<u32*>(small.as<u8>() + 3588));
            return 0;
        }
        int rc = directory_layout(n);
        if (rc) return rc;
        const int d = q_dbits;
        const u64 nb = (1ull << d) + 1;
        const int cbits = dir_coarse_bits(d);
        const u64 ncb = cbits ? (1ull << cbits) + 1 : 0;
        u32* fine = qdir.as<u32>();
        u32* coarse = cbits ? fine + nb : nullptr;
        if (cbits)
            hipLaunchKernelGGL(dir_build_kernel, dim3(stream_grid(ncb, 256)), dim3(256), 0, stream, qkeys, n, cbits, coarse,
                               (const u32*)nullptr, 0);
        hipLaunchKernelGGL(dir_build_kernel, dim3(stream_grid(nb, 256)), dim3(256), 0, stream, qkeys, n, d, fine, (const u32*)coarse, cbits);
        SA_HIP_CHECK(hipGetLastError());
        return 0;
    }

    // Query structures for an adopted (text, SA): packed keys gathered from the text.
    int prepare_query_from_sa(const CodeMap& map, int b, u32 L) {
        qkeys = nullptr;
        qkeys32 = nullptr;
        dir_ready = false;
        k2_ready = false;
        if (n < 2) return 0;
        const int k0 = choose_initial_chars(b, L);
        int rc = keys0.ensure((size_t)n * 8 + 64);
        if (rc) return rc;
        hipLaunchKernelGGL(gather_keys_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, stream, text.as<u8>(), n, map, b, k0,
                           (const u32*)sa, keys0.as<u64>());
        qkeys = keys0.as<u64>();
        qmap = map; q_b = b; q_k0 = k0;
        return build_directory();
    }

    // Number of violations of the suffix-array property (0 = verified).
    int verify(u64* violations) {
        *violations = 0;
        if (n == 0) return 0;
        int rc = isa.ensure((size_t)n * 4);
        if (rc) return rc;
        u64* bad = reinterpret_cast<u64*>(small.as<u8>() + 3072);
        SA_HIP_CHECK(hipMemsetAsync(bad, 0, 8, stream));
        SA_HIP_CHECK(hipMemsetAsync(isa.p, 0xFF, (size_t)n * 4, stream));
        const u32 g = stream_grid(n, 1024);
        hipLaunchKernelGGL(verify_scatter_kernel, dim3(g), dim3(256), 0, stream, (const u32*)sa, n, isa.as<u32>(), bad);
        if (max_suffix_length == 0)
            hipLaunchKernelGGL(verify_order_kernel, dim3(g), dim3(256), 0, stream, text.as<u8>(), (const u32*)sa, isa.as<u32>(), n, bad);
        else
            hipLaunchKernelGGL(verify_truncated_kernel, dim3(g), dim3(256), 0, stream, text.as<u8>(), (const u32*)sa, isa.as<u32>(), n,
                               max_suffix_length, bad);
        SA_HIP_CHECK(hipMemcpyAsync(violations, bad, 8, hipMemcpyDeviceToHost, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        return 0;
    }

    // Sort of one round's records (keys = gid << gid_shift | low bits, ordered by gid already; values = suffix indices):
    // groups are sorted where they lie, tile by tile in LDS (round_sort.hpp); the few groups that do not fit a tile go
    // through the global sort as a compact list.  Falls back to the global sort of everything when more than half of
    // the records sit in such groups (long repeats: a few huge groups).  Result in (k1, v1) or wherever the global sort
    // leaves it.
    // vsrc: the values in list order (read only; the global sort works on a copy of them in v0).
    int round_sort(u64* k0, const u32* vsrc, u32* v0, u64* k1, u32* v1, u32 M, u32 G, int begin_bit, int end_bit, int gid_shift,
                   u64** kres, u32** vres) {
        int rc;
        auto global_sort = [&]() -> int {
            SA_HIP_CHECK(hipMemcpyAsync(v0, vsrc, (size_t)M * 4, hipMemcpyDeviceToDevice, stream));
            return radix_sort_pairs(radix, stream, k0, v0, k1, v1, M, begin_bit, end_bit, false, false, kres, vres);
        };
        int top = (gid_shift < 64 - LOC_GID_BITS) ? gid_shift + LOC_GID_BITS : 64;
        if (top > end_bit) top = end_bit;
        // (an average group of more than half a tile: most records would take the big-group route anyway -- all-'a',
        //  Fibonacci strings: no planning, no extra synchronisation)
        if (!local_rounds || G == 0 || top <= begin_bit || (M > LOC_CAP && (u64)M > (u64)G * (LOC_CAP / 2)))
            return global_sort();
        u32 ntiles = div_up(M, LOC_TILE);
        if ((rc = gstart.ensure(((size_t)G + 2) * 4))) return rc;
        if ((rc = loc_tiles.ensure((size_t)ntiles * sizeof(LocTile) + 64))) return rc;
        u32* total_dev = reinterpret_cast<u32*>(small.as<u8>() + 3616);
        u32 big = 0;
        if (M <= LOC_CAP) {
            // the whole list fits one workgroup (the last rounds of a build): one tile, nothing to plan, no round trip
            one_tile.begin = 0; one_tile.local_end = M; one_tile.end = M; one_tile.big_off = 0;
            SA_HIP_CHECK(hipMemcpyAsync(loc_tiles.p, &one_tile, sizeof one_tile, hipMemcpyHostToDevice, stream));
            ntiles = 1;
        } else {
        hipLaunchKernelGGL(group_starts_kernel, dim3(stream_grid(M, 1024)), dim3(256), 0, stream, gid.as<u32>(), M, G, gstart.as<u32>());
        hipLaunchKernelGGL((loc_plan_kernel<LOC_TILE, LOC_CAP>), dim3(div_up(ntiles, 256)), dim3(256), 0, stream, gid.as<u32>(), gstart.as<u32>(), M, ntiles,
                           loc_tiles.as<LocTile>());
        hipLaunchKernelGGL(loc_scan_kernel, dim3(1), dim3(1024), 0, stream, loc_tiles.as<LocTile>(), ntiles, total_dev);
        SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_BIG, total_dev, 4, hipMemcpyDeviceToHost, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        memcpy(&big, mbox + MB_BIG, 4);
        if ((u64)big * 2 > M)
            return global_sort();
        }
        LocSortArgs a;
        a.keys_in = k0; a.vals_in = vsrc; a.keys_out = k1; a.vals_out = v1; a.tiles = loc_tiles.as<LocTile>();
        a.begin_bit = begin_bit; a.gid_shift = gid_shift; a.top = top;
        a.passes = (top - begin_bit + RADIX_BITS - 1) / RADIX_BITS;
        if (top - begin_bit + LOC_GID_BITS <= 64) hipLaunchKernelGGL(loc_sort_kernel<true>, dim3(ntiles), dim3(LOC_BLOCK), 0, stream, a);
        else hipLaunchKernelGGL(loc_sort_kernel<false>, dim3(ntiles), dim3(LOC_BLOCK), 0, stream, a);
        local_records += (u64)M - big;
        if (big) {
            big_records += big;
            if ((rc = big_keys.ensure((size_t)big * 8))) return rc;
            if ((rc = big_vals.ensure((size_t)big * 4))) return rc;
            const u32 cg = ntiles < 2048u ? ntiles : 2048u;
            hipLaunchKernelGGL(loc_big_copy_kernel, dim3(cg, 16), dim3(256), 0, stream, (const LocTile*)loc_tiles.as<LocTile>(), ntiles, true,
                               k0, const_cast<u32*>(vsrc), big_keys.as<u64>(), big_vals.as<u32>());
            // (k0, v0) are free from here on: the partner buffers of the list's ping-pong
            u64* rk; u32* rv;
            if ((rc = radix_sort_pairs(radix, stream, big_keys.as<u64>(), big_vals.as<u32>(), k0, v0, big, begin_bit, end_bit, false,
                                       false, &rk, &rv))) return rc;
            hipLaunchKernelGGL(loc_big_copy_kernel, dim3(cg, 16), dim3(256), 0, stream, (const LocTile*)loc_tiles.as<LocTile>(), ntiles, false,
                               k1, v1, rk, rv);
        }
        SA_HIP_CHECK(hipGetLastError());
        *kres = k1;
        *vres = v1;
        return 0;
    }

    // Groups that fit a tile are refined to the end in LDS (group_finish.hpp); what it resolves leaves the active list.
    // In: the active list (apos_cur, aidx, gid) of M records in G groups at depth h.  Out: M, G, the lists compacted.
    int run_group_finisher(const CodeMap& map, int b, u32 L, u64 h, u32& M, u32& G, u32* tot) {
        u32*& apos_cur = lst_cur;
        u32*& apos_nxt = lst_nxt;
        int rc;
        u32 ntiles = div_up(M, FIN_TILE);
        if ((rc = gstart.ensure(((size_t)G + 2) * 4))) return rc;
        if ((rc = loc_tiles.ensure((size_t)ntiles * sizeof(LocTile) + 64))) return rc;
        if ((rc = done.ensure((size_t)M + 64))) return rc;
        if ((rc = fin_flag.ensure((size_t)M + 64))) return rc;
        if (M <= FIN_CAP) {
            one_tile.begin = 0; one_tile.local_end = M; one_tile.end = M; one_tile.big_off = 0;
            SA_HIP_CHECK(hipMemcpyAsync(loc_tiles.p, &one_tile, sizeof one_tile, hipMemcpyHostToDevice, stream));
            ntiles = 1;
        } else {
            hipLaunchKernelGGL(group_starts_kernel, dim3(stream_grid(M, 1024)), dim3(256), 0, stream, gid.as<u32>(), M, G, gstart.as<u32>());
            hipLaunchKernelGGL((loc_plan_kernel<FIN_TILE, FIN_CAP>), dim3(div_up(ntiles, 256)), dim3(256), 0, stream, gid.as<u32>(), gstart.as<u32>(), M,
                               ntiles, loc_tiles.as<LocTile>());
        }
        unsigned long long* ft = reinterpret_cast<unsigned long long*>(small.as<u8>() + 3840);
        SA_HIP_CHECK(hipMemsetAsync(ft, 0, 256, stream));
        SA_HIP_CHECK(hipMemsetAsync(done.p, 0, (size_t)M, stream));   // records of groups too large for a tile stay untouched
        FinArgs a;
        a.text = text.as<u8>(); a.n = n; a.b = b;
        a.aidx = aidx.as<u32>(); a.gid = gid.as<u32>(); a.apos = apos_cur; a.tiles = loc_tiles.as<LocTile>();
        a.h0 = (u32)h; a.L = L; a.max_rounds = FIN_MAX_ROUNDS;
        a.count_max = fin_count_max; a.radix_chars = fin_radix_chars; a.debug = debug_rounds ? 1 : 0;
        a.sa = sa; a.gflags = flags.as<u8>(); a.done = done.as<u8>();
        a.res_idx = ridx0.as<u32>(); a.res_fin = fin_flag.as<u8>(); a.totals = ft;
        if (fin_v2) {
            // (ridx0 = the list's ping-pong partner, free here: 4 bytes per record; the fetched words need 8)
            a.w0 = nullptr;
            if (fin_prefetch) {
                if ((rc = fin_w0.ensure((size_t)M * 8 + 64))) return rc;
                a.w0 = fin_w0.as<u64>();
                hipLaunchKernelGGL(fin_prefetch_kernel, dim3(ntiles), dim3(256), 0, stream, (const u8*)text.as<u8>(), (const u32*)aidx.as<u32>(),
                                   (const LocTile*)loc_tiles.as<LocTile>(), (u32)h, fin_w0.as<u64>());
            }
            hipLaunchKernelGGL(group_finish2_kernel, dim3(ntiles), dim3(FIN_BLOCK), 0, stream, a, map);
        }
        else hipLaunchKernelGGL(group_finish_kernel, dim3(ntiles), dim3(FIN_BLOCK), 0, stream, a, map);
        unsigned long long ft_host[32] = {0};
        SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_FT, ft, 256, hipMemcpyDeviceToHost, stream));
        // What the run leaves.  When no tile gave up (every record inside the tiles was resolved: totals[0] == totals[1]) that is
        // exactly the groups the plan left out -- copied to the next lists, speculatively, before the totals are known: one
        // synchronisation either way.  Otherwise (long repeats) the done flags decide, as before.
        bool left_fast = false;
        u32 left_host[2] = {0, 0};
        if (fin_left_fast && ntiles > 1 && !fin_left_failed) {
            // the sizes of what the plan left out (one small workgroup) travel with the finisher's totals: one synchronisation;
            // the copy itself only when it is known to be the whole story
            u32* left_tot = reinterpret_cast<u32*>(small.as<u8>() + 3632);
            if ((rc = fin_left_gid.ensure((size_t)ntiles * 4 + 64))) return rc;
            hipLaunchKernelGGL(left_scan_kernel, dim3(1), dim3(1024), 0, stream, loc_tiles.as<LocTile>(), ntiles, fin_left_gid.as<u32>(), left_tot);
            SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_BIG, left_tot, 8, hipMemcpyDeviceToHost, stream));
            SA_HIP_CHECK(hipStreamSynchronize(stream));
            memcpy(ft_host, mbox + MB_FT, 256);
            memcpy(left_host, mbox + MB_BIG, 8);
            left_fast = ft_host[0] == ft_host[1] && (u64)ft_host[0] + left_host[0] == M;
            if (left_fast && left_host[0]) {
                const u32 cg = ntiles < 2048u ? ntiles : 2048u;
                // (ridx0 was the kernel's scratch and is free again; ridx1 holds 4 bytes per record of the first list like ridx0)
                hipLaunchKernelGGL(left_copy_kernel, dim3(cg, 16), dim3(256), 0, stream, (const LocTile*)loc_tiles.as<LocTile>(), ntiles,
                                   (const u32*)fin_left_gid.as<u32>(), (const u32*)apos_cur, (const u32*)aidx.as<u32>(), apos_nxt, ridx0.as<u32>(), ridx1.as<u32>());
            }
            if (!left_fast) fin_left_failed = true;   // a text whose groups defeat the finisher (long repeats) does so in every run: no second attempt in this build
        }
        if (left_fast) {
            tot[0] = left_host[0]; tot[1] = left_host[1];
            if (tot[0]) {
                SA_HIP_CHECK(hipMemcpyAsync(aidx.p, ridx0.p, (size_t)tot[0] * 4, hipMemcpyDeviceToDevice, stream));
                SA_HIP_CHECK(hipMemcpyAsync(gid.p, ridx1.p, (size_t)tot[0] * 4, hipMemcpyDeviceToDevice, stream));
                swap_lists();
            }
            stats.finisher_runs += 1;
            stats.finisher_records += ft_host[0];
            stats.finisher_resolved += (u64)M - tot[0];
            if (ft_host[1] * 4 < ft_host[0]) fin_useful = false;
            M = tot[0];
            G = tot[1];
            return 0;
        }
        const u32 tiles = div_up(M, BLD_TILE);
        hipLaunchKernelGGL(tiny_flags_kernel, dim3(tiles), dim3(BLD_BLOCK), 0, stream, gid.as<u32>(), done.as<u8>(), M, lf.as<u8>(),
                           counts.as<uint2>());
        if ((rc = scan_counts(tiles, tot))) return rc;   // synchronises: the mailbox holds the finisher's totals
        memcpy(ft_host, mbox + MB_FT, 256);
        stats.finisher_runs += 1;
        stats.finisher_records += ft_host[0];
        stats.finisher_resolved += (u64)M - tot[0];
        if (ft_host[1] * 4 < ft_host[0]) fin_useful = false;   // long repeats: leave them to the doubling rounds
        if (debug_rounds)
            fprintf(stderr, "[sa_hip] finisher h=%llu M=%u G=%u tiles=%u (%llu non-empty): looked at %llu, resolved %llu -> M'=%u G'=%u; rounds %llu (%llu radix), "
                    "slots radix %llu counting %llu, active record-rounds %llu\n", (unsigned long long)h, M, G, ntiles, ft_host[6], ft_host[0], ft_host[1],
                    tot[0], tot[1], ft_host[2], ft_host[3], ft_host[4], ft_host[5], ft_host[7]);
        if (debug_rounds && ft_host[6])
            fprintf(stderr, fin_v2 ? "[sa_hip]   finisher v2 (rounds with a split group %llu; split records seen by wave 0: %llu) cycles per tile (thread 0, clock64): fetch+keys %llu, counting + splits %llu, permutation %llu, regroup + finals %llu, tail %llu\n"
                                   : "[sa_hip]   finisher (radix rounds %llu, radix slots %llu) cycles per tile (thread 0, clock64): fetch+keys %llu, counting sort %llu, radix sort %llu, regroup %llu, write-out %llu\n",
                    ft_host[3], ft_host[4], ft_host[8] / ft_host[6], ft_host[9] / ft_host[6], ft_host[10] / ft_host[6], ft_host[11] / ft_host[6], ft_host[12] / ft_host[6]);
        if (debug_rounds && ft_host[6]) {
            fprintf(stderr, "[sa_hip]   records by group size at entry [2^c, 2^(c+1)):");
            for (int c = 1; c < 13; ++c) fprintf(stderr, " %d:%llu", c, ft_host[16 + c]);
            fprintf(stderr, "\n");
        }
        if (tot[0] < M) {
            if (tot[0]) {
                SA_HIP_CHECK(hipMemcpyAsync(ridx0.p, aidx.p, (size_t)M * 4, hipMemcpyDeviceToDevice, stream));
                launch_compact(lf.as<u8>(), M, tot[0], apos_cur, ridx0.as<u32>(), apos_nxt, aidx.as<u32>(), gid.as<u32>());
                swap_lists();
            }
            M = tot[0];
            G = tot[1];
        }
        return 0;
    }

    // Long repeats (period_finish.hpp): tied groups whose members form an arithmetic progression inside one periodic run are
    // ordered by one comparison.  In: the active list (lst_cur, aidx, gid) of M records in G groups.  Out: M, G, lists compacted.
    int run_period_finisher(bool have_isa, u32& M, u32& G, u32* tot) {
        int rc;
        if ((rc = gstart.ensure(((size_t)G + 2) * 4))) return rc;
        if ((rc = per_gd.ensure((size_t)G * 4 + 64))) return rc;
        if ((rc = per_bad.ensure((size_t)G + 64))) return rc;
        if ((rc = per_dec.ensure((size_t)G + 64))) return rc;
        if ((rc = per_table.ensure((size_t)PER_TABLE * sizeof(uint2)))) return rc;
        if ((rc = done.ensure((size_t)M + 64))) return rc;
        const u32 ntiles = div_up(n, PER_TILE);
        if ((rc = per_tf.ensure((size_t)ntiles * 4 + 64))) return rc;
        if ((rc = per_carry.ensure((size_t)ntiles * 4 + 64))) return rc;
        SA_HIP_CHECK(hipMemsetAsync(done.p, 0, (size_t)M, stream));
        hipLaunchKernelGGL(group_starts_kernel, dim3(stream_grid(M, 1024)), dim3(256), 0, stream, gid.as<u32>(), M, G, gstart.as<u32>());
        hipLaunchKernelGGL(per_init_kernel, dim3(stream_grid(G, 256)), dim3(256), 0, stream, aidx.as<u32>(), gstart.as<u32>(), G,
                           per_gd.as<u32>(), per_bad.as<u8>());
        hipLaunchKernelGGL(per_classify_kernel, dim3(stream_grid(M, 1024)), dim3(256), 0, stream, aidx.as<u32>(), gid.as<u32>(), gstart.as<u32>(), M,
                           per_gd.as<u32>(), per_bad.as<u8>());
        u32* best_dev = reinterpret_cast<u32*>(small.as<u8>() + 3720);
        u64 covered = 0;
        for (int iter = 0; iter < 4; ++iter) {
            SA_HIP_CHECK(hipMemsetAsync(per_table.p, 0, (size_t)PER_TABLE * sizeof(uint2), stream));
            hipLaunchKernelGGL(per_hist_kernel, dim3(stream_grid(G, 256)), dim3(256), 0, stream, gstart.as<u32>(), G, per_gd.as<u32>(),
                               per_bad.as<u8>(), per_table.as<uint2>());
            hipLaunchKernelGGL(per_pick_kernel, dim3(1), dim3(1024), 0, stream, per_table.as<uint2>(), best_dev);
            u32 best[2] = {0, 0};
            SA_HIP_CHECK(hipMemcpyAsync(mbox + MB_BEST, best_dev, 8, hipMemcpyDeviceToHost, stream));
            SA_HIP_CHECK(hipStreamSynchronize(stream));
            memcpy(best, mbox + MB_BEST, 8);
            if (debug_rounds) fprintf(stderr, "[sa_hip] period finisher M=%u G=%u: difference %u covers %u records\n", M, G, best[0], best[1]);
            if (best[0] == 0 || (u64)best[1] * 8 < M) break;   // nothing periodic enough (left)
            covered += best[1];
            const u64 d = best[0];
            hipLaunchKernelGGL(per_tile_first_kernel, dim3(ntiles), dim3(256), 0, stream, text.as<u8>(), n, d, per_tf.as<u32>());
            hipLaunchKernelGGL(per_tile_scan_kernel, dim3(1), dim3(1024), 0, stream, per_tf.as<u32>(), ntiles, per_carry.as<u32>());
            PerArgs a;
            a.text = text.as<u8>(); a.n = n; a.d = d;
            a.aidx = aidx.as<u32>(); a.apos = lst_cur; a.gid = gid.as<u32>(); a.gstart = gstart.as<u32>(); a.gd = per_gd.as<u32>();
            a.bad = per_bad.as<u8>(); a.G = G; a.M = M; a.tile_first = per_tf.as<u32>(); a.carry = per_carry.as<u32>(); a.dec = per_dec.as<u8>();
            a.sa = sa; a.gflags = flags.as<u8>(); a.done = done.as<u8>(); a.isa = have_isa ? isa.as<u32>() : nullptr;
            hipLaunchKernelGGL(per_decide_kernel, dim3(stream_grid((u64)G * 64, 256)), dim3(256), 0, stream, a);
            hipLaunchKernelGGL(per_apply_kernel, dim3(stream_grid(M, 1024)), dim3(256), 0, stream, a);
        }
        if ((u64)covered * 8 < M) { per_skip = 2 << per_fails++; return 0; }   // nothing was attempted: the lists are untouched
        const u32 tiles = div_up(M, BLD_TILE);
        hipLaunchKernelGGL(tiny_flags_kernel, dim3(tiles), dim3(BLD_BLOCK), 0, stream, gid.as<u32>(), done.as<u8>(), M, lf.as<u8>(),
                           counts.as<uint2>());
        if ((rc = scan_counts(tiles, tot))) return rc;
        stats.period_resolved += (u64)M - tot[0];
        if (debug_rounds) fprintf(stderr, "[sa_hip] period finisher: resolved %u of %u -> M'=%u G'=%u\n", M - tot[0], M, tot[0], tot[1]);
        if ((u64)(M - tot[0]) * 8 < M) per_skip = 2 << per_fails++;   // (chains cut by the end of a run split up in later rounds)
        if (tot[0] < M) {
            if (tot[0]) {
                SA_HIP_CHECK(hipMemcpyAsync(ridx0.p, aidx.p, (size_t)M * 4, hipMemcpyDeviceToDevice, stream));
                launch_compact(lf.as<u8>(), M, tot[0], lst_cur, ridx0.as<u32>(), lst_nxt, aidx.as<u32>(), gid.as<u32>());
                swap_lists();
            }
            M = tot[0];
            G = tot[1];
        }
        return 0;
    }

    // The device build.  Text already resident in text.p[0..n_).
    // sa64_out (may be null): the suffix array also in libsais64 layout, int64[n] in a device buffer of the caller.
    // On the narrow-record plan the sort's last pass writes it next to the u32 array and the slots refined afterwards
    // are patched from the first active list; otherwise one widening pass at the end.
    int build(u64 n_, u32 L, int64_t* sa64_out = nullptr) {
        int rc;
        memset(&stats, 0, sizeof stats);
        local_records = big_records = 0;
        fin_useful = true;
        fin_left_failed = false;
        per_skip = per_fails = 0;
        radix.reset_stats();
        max_suffix_length = L;
        stats.n = n_;
        if (n_ > n_max) return fail(SA_HIP_EINVAL, "text longer than the index capacity");
        if ((rc = ensure_build_buffers())) return rc;
        SA_HIP_CHECK(hipEventRecord(ev_begin, stream));
        CodeMap map;
        u32 sigma = 0;
        int b = 1;
        if ((rc = prepare_text(n_, map, sigma, b))) return rc;
        stats.sigma = sigma;
        stats.bits_per_symbol = (u32)b;
        sa = vals0.as<u32>();
        qkeys = nullptr;
        qkeys32 = nullptr;
        dir_ready = false;
        dir_by_local_pass = false;
        k2_ready = false;
        stats.widen_fused = 0;
        if (n == 0) return finish_stats();
        if (n == 1) {
            SA_HIP_CHECK(hipMemsetAsync(sa, 0, 4, stream));
            SA_HIP_CHECK(hipMemsetAsync(flags.p, 1, 1, stream));
            if (sa64_out) SA_HIP_CHECK(hipMemsetAsync(sa64_out, 0, 8, stream));
            return finish_stats();
        }
        const int k0 = choose_initial_chars(b, L);
        stats.initial_chars = (u32)k0;
        const u32 n32 = (u32)n;

        // initial keys (+ fused digit histograms) + sort #0
        SortPlan pl;
        const int begin_bit = 64 - b * k0;
        if ((rc = make_plan(radix, n32, begin_bit, 64, pl))) return rc;
        const bool narrow_path = narrow_sort && fuse_hist && narrow_sort_applies(radix, n, begin_bit);
        const bool narrow48_path = !narrow_path && narrow48_possible(b) && narrow48_applies(radix, n, begin_bit, b, k0);
        stats.narrow48 = narrow48_path ? 1u : 0u;
        if (fuse_hist) { if ((rc = radix_prepare(radix, stream))) return rc; }
        // the narrow sort starts with the TOP digit, the plain LSD sort with the lowest one; when the top-digit
        // pass reads the text itself, key generation shrinks to the histogram of that digit
        const bool text_pass = narrow_path && text_top_pass && text_pass_applies(b, k0);
        stats.text_top_pass = text_pass ? 1u : 0u;
        // (a narrow sort fed from a u64 key array -- no text pass -- needs the wide buffers for that array)
        if ((rc = ensure_key_buffers(n, narrow_path && text_pass && narrow_k && fuse_directory))) return rc;
        if (diag_env("SA_HIP_DEBUG_ADDR"))   // diagnostic: where the streams of the sort passes lie (tools/gpu_addr.sh)
            fprintf(stderr, "[sa_hip] addr text=%p keys0=%p keys1=%p vals0=%p vals1=%p sa64=%p\n", text.p, keys0.p, keys1.p, vals0.p, vals1.p, (void*)sa64_out);
        // the plain 12-byte-record sort can take its pass 0 from the text too (no key array written and read back)
        const bool wide_text = !narrow_path && !narrow48_path && wide_text_pass && fuse_hist && radix.block == 512 && text_pass_applies(b, k0) &&
                               pl.npasses > 1 && n >= (1u << 16);
        stats.text_top_pass = (text_pass || wide_text || narrow48_path) ? 1u : 0u;
        if (text_pass || narrow48_path) {
            if ((rc = narrow_text_histogram(radix, narrow, stream, text.as<u8>(), map, n32, b))) return rc;
        } else if (wide_text) {
            if ((rc = wide_text_histogram(radix, narrow, stream, text.as<u8>(), map, n32, b, k0))) return rc;
        } else {
            hipLaunchKernelGGL(keygen_kernel, dim3(stream_grid(n, BLD_TILE)), dim3(BLD_BLOCK), 0, stream, text.as<u8>(), n, map, b,
                               k0, keys0.as<u64>(), pl.g, narrow_path ? 56 : pl.shift(0), narrow_path ? 255u : pl.mask(0),
                               fuse_hist ? radix.hist(0) : (u32*)nullptr);
        }
        u64* kres; u32* vres;
        // the narrow keys can stay narrow when the query directory comes out of the flags pass below (the directory
        // by binary search reads u64 keys)
        const bool keep_narrow = narrow_path && narrow_k && fuse_directory;
        stats.narrow_k = keep_narrow ? 1u : 0u;
        if (narrow_path) {
            TextSource src;
            src.text = text.as<u8>(); src.b = b; src.k0 = k0;
            LocalFlagsRequest freq;   // the first flags pass inside the sort's last pass, when the three-pass plan is taken
            if (keep_narrow && lite_flags) { freq.prepare = &Builder::split_flags_prepare; freq.ctx = this; }
            if ((rc = radix_sort_narrow(radix, narrow, stream, keys0.as<u64>(), vals0.as<u32>(), keys1.as<u64>(), vals1.as<u32>(),
                                        n32, begin_bit, &kres, &vres, text_pass ? &src : nullptr, keep_narrow, sa64_out, &freq))) return rc;
            if (sa64_out) stats.widen_fused = 1;
            stats.split_plan = narrow.split_used ? (u32)narrow.split_rb : 0u;
            stats.split_max = narrow.split_max_seen;
        } else if (narrow48_path) {
            // The 56 key bits hold k0 whole characters and, when the next one does not fit, its TOP bits: the key is the first
            // 56 bits of the (k0 + 1)-character key.  Groups are still classes of the first k0 characters (h = k0) cut a little
            // finer -- in an order consistent with the suffix order -- so fewer suffixes reach the finisher, for free.
            TextSource src;
            src.text = text.as<u8>(); src.b = b; src.k0 = k0;
            int begin48 = begin_bit;
            if (partial_char && b * k0 < 56 && b * (k0 + 1) > 56 && (L == 0 || (u32)(k0 + 1) <= L) && text_pass_applies(b, k0 + 1)) {
                src.k0 = k0 + 1;
                begin48 = 8;
            }
            if ((rc = radix_sort_narrow48(radix, narrow, stream, keys0.as<u64>(), vals0.as<u32>(), keys1.as<u64>(), vals1.as<u32>(),
                                          n32, begin48, &kres, &vres, src, sa64_out))) return rc;
            if (sa64_out) stats.widen_fused = 1;
        } else {
            WideTextCtx wctx{text.as<u8>(), narrow.map_dev, n, b, k0};
            Pass0Launcher p0;
            p0.launch = &launch_text_low_pass; p0.ctx = &wctx; p0.bytes_per_record = 13;
            if ((rc = radix_sort_pairs(radix, stream, keys0.as<u64>(), vals0.as<u32>(), keys1.as<u64>(), vals1.as<u32>(), n32,
                                       begin_bit, 64, true, fuse_hist, &kres, &vres, wide_text ? &p0 : nullptr))) return rc;
        }
        sa = vres;
        // sorted packed keys: kept for the query path (sa_query.hpp)
        NarrowKeys nk{};
        if (keep_narrow) {
            qkeys32 = reinterpret_cast<const u32*>(kres);
            q_bstart = narrow.plan->bstart;
            q_lo_shift = begin_bit;
            nk.keys32 = qkeys32; nk.bstart = q_bstart; nk.lo_shift = q_lo_shift;
        } else qkeys = kres;
        qmap = map; q_b = b; q_k0 = k0;

        // head flags, active counts
        u32 tot[2];
        flags_valid = true;
        bool staged = false;   // the active records sit in the lite pass's staging rows
        if (keep_narrow && lite_flags) {
            // near-random text: the lite form of the pass (no flag array; the active records staged in the key buffer the sort
            // has left free), the full form only when a tile overflows its staging row
            void* stage = (static_cast<void*>(kres) == keys0.p) ? keys1.p : keys0.p;
            bool overflow = false;
            if (narrow_path && narrow.split_flags_done) { if ((rc = flags_after_split(nk, tot, &overflow))) return rc; }
            else
            if ((rc = flags_lite_pass(nk, n32, stage, tot, &overflow))) return rc;
            if (overflow) { if ((rc = flags_and_counts(kres, n32, flags.as<u8>(), nullptr, nullptr, tot, false, &nk))) return rc; }
            else { flags_valid = false; staged = true; }
            stats.lite_flags = staged ? ((narrow_path && narrow.split_flags_done) ? 2u : 1u) : 0u;
        } else
        if ((rc = flags_and_counts(kres, n32, flags.as<u8>(), nullptr, nullptr, tot, fuse_directory, keep_narrow ? &nk : nullptr))) return rc;
        if ((rc = check_device_status())) return rc;
        u32 M = tot[0], G = tot[1];
        u64 h = (u64)k0;
        bool have_isa = false;
        int chunk_done = 0;

        if (M && (L == 0 || h < L)) {
            const size_t m0 = M;
            if ((rc = apos0.ensure(m0 * 4))) return rc;
            if ((rc = apos1.ensure(m0 * 4))) return rc;
            if ((rc = aidx.ensure(m0 * 4))) return rc;
            if ((rc = gid.ensure(m0 * 4))) return rc;
            if ((rc = rkeys0.ensure(m0 * 8))) return rc;
            if ((rc = rkeys1.ensure(m0 * 8))) return rc;
            if ((rc = ridx0.ensure(m0 * 4))) return rc;
            if ((rc = ridx1.ensure(m0 * 4))) return rc;
            if ((rc = lf.ensure(m0 + 64))) return rc;
            // first compaction: domain = whole SA
            if (staged)
                hipLaunchKernelGGL(lite_gather_kernel, dim3(div_up(lite_tiles, 4)), dim3(256), 0, stream, (const uint2*)counts.as<uint2>(),
                                   lite_tiles, (const u32*)totals_dev(), lite, apos0.as<u32>(), aidx.as<u32>(), gid.as<u32>());
            else
            launch_compact(flags.as<u8>(), n32, M, nullptr, sa, apos0.as<u32>(), aidx.as<u32>(), gid.as<u32>());
        }
        const u32 M0 = (M && (L == 0 || h < L)) ? M : 0;   // the first active list (apos0) stays: the slots an int64 copy has to be patched at
        bool patched_by_tiny = false;
        lst_cur = apos0.as<u32>();
        lst_nxt = apos1.as<u32>();
        lst_first = true;
        lst_keep_first = M0 && stats.widen_fused;
        if (lst_keep_first) {
            if ((rc = apos2.ensure((size_t)M0 * 4))) return rc;
        }
        u32*& apos_cur = lst_cur;
        u32*& apos_nxt = lst_nxt;
        const int rb = bits_for(n + 1);

        // tiny-group finisher: pairs .. octets of tied suffixes are ordered by direct comparison
        // (only for sparse active sets: there the alternative is a launch-bound 8-pass round over a few
        //  records; dense active sets are served better by the bandwidth-bound radix rounds)
        if (tiny_finisher && M && (u64)M * 16 <= n && (L == 0 || h < L)) {
            if ((rc = done.ensure((size_t)M + 64))) return rc;
            SA_HIP_CHECK(hipMemsetAsync(done.p, 0, (size_t)M, stream));
            const u32 limit = L ? (u32)(L - h) : TINY_DEPTH;
            hipLaunchKernelGGL(tiny_groups_kernel, dim3(stream_grid(M, 256)), dim3(256), 0, stream, text.as<u8>(), n, aidx.as<u32>(),
                               gid.as<u32>(), apos_cur, M, h, limit, L ? 1 : 0, sa, flags_valid ? flags.as<u8>() : (u8*)nullptr, done.as<u8>(),
                               (sa64_out && stats.widen_fused) ? sa64_out : (int64_t*)nullptr);
            const u32 tiles = div_up(M, BLD_TILE);
            hipLaunchKernelGGL(tiny_flags_kernel, dim3(tiles), dim3(BLD_BLOCK), 0, stream, gid.as<u32>(), done.as<u8>(), M,
                               lf.as<u8>(), counts.as<uint2>());
            if ((rc = scan_counts(tiles, tot))) return rc;
            stats.tiny_resolved = (u64)M - tot[0];
            if (tot[0] == 0 && M == M0) patched_by_tiny = true;   // every slot the int64 copy lacked has just been written in both widths
            if (tot[0] && !flags_valid) {
                // something is left for the rounds, which mark heads in the flag array: build it now, with the marks the tiny pass
                // would have made
                if ((rc = materialise_flags())) return rc;
                hipLaunchKernelGGL(mark_done_heads_kernel, dim3(stream_grid(M, 1024)), dim3(256), 0, stream, (const u32*)apos_cur, (const u8*)done.as<u8>(), M,
                                   flags.as<u8>());
            }
            if (tot[0] < M) {
                if (tot[0]) {
                    SA_HIP_CHECK(hipMemcpyAsync(ridx0.p, aidx.p, (size_t)M * 4, hipMemcpyDeviceToDevice, stream));
                    launch_compact(lf.as<u8>(), M, tot[0], apos_cur, ridx0.as<u32>(), apos_nxt, aidx.as<u32>(), gid.as<u32>());
                    swap_lists();
                }
                M = tot[0];
                G = tot[1];
            }
        }

        if (M && (L == 0 || h < L) && (rc = materialise_flags())) return rc;   // (no-op unless the lite pass ran and no tiny pass followed)
        u32 M_prev = 0;   // active-set size of the previous round (0: none yet)
        while (M && (L == 0 || h < L)) {
            // groups that fit a tile are finished in LDS, whatever their number of rounds; the global round below is for
            // the rest (an average group of more than half a tile: hardly anything fits)
            bool finisher_ran = false;
            if (group_finish && fin_useful && !have_isa && (u64)M <= (u64)G * (FIN_CAP / 2)) {
                if ((rc = run_group_finisher(map, b, L, h, M, G, tot))) return rc;
                if (!M) break;
                finisher_ran = fin_useful;
            }
            // long repeats -- the finisher gave up, doubling is under way, or the groups are far larger than a tile: groups that
            // are arithmetic progressions inside one periodic run are ordered in one step (full suffix arrays only)
            if (period_finish && L == 0 && M >= 64 && per_fails < 8 && (!fin_useful || have_isa || (u64)M > (u64)G * (FIN_CAP / 2))) {
                if (per_skip > 0) --per_skip;
                else {
                    if ((rc = run_period_finisher(have_isa, M, G, tot))) return rc;
                    if (!M) break;
                }
            }
            const int gb = bits_for(G);
            // Chunk rounds read the next characters from the text; doubling rounds need the inverse suffix array
            // (one random 4-byte write per suffix of the WHOLE text the first time) and pay off only on long
            // repeats.  As long as the active set at least halves from round to round the remaining chunk rounds
            // are cheaper than that scatter (words / names: 60.8M -> 20.5M -> 4.2M -> 0.5M -> 33K -> 633 -> 12), so
            // doubling starts only when the set has stopped shrinking after the first chunk rounds.
            const bool shrinking = M_prev != 0 && (u64)M * 2 <= (u64)M_prev;
            bool use_chunk = (L != 0) || (chunk_done < chunk_rounds_before_doubling) || (adaptive_doubling && !have_isa && shrinking);
            int kc = 0;
            if (use_chunk) {
                kc = (64 - gb) / b;
                // what the finisher left are groups too large for a tile: a few characters take them apart far enough for
                // the next finisher run, and every 8 key bits less is a pass less over their records
                if (finisher_ran && big_round_chars > 0 && kc > big_round_chars) kc = big_round_chars;
                if (L && (u64)kc > L - h) kc = (int)(L - h);
                if (kc <= 0) use_chunk = false;
            }
            int begin_bit;
            u64 h_next;
            if (use_chunk) {
                hipLaunchKernelGGL(chunk_keys_kernel, dim3(stream_grid(M, 256)), dim3(256), 0, stream, text.as<u8>(), n, map, b,
                                   aidx.as<u32>(), gid.as<u32>(), M, (u32)h, kc, gb, rkeys0.as<u64>());
                begin_bit = 64 - gb - b * kc;
                h_next = h + kc;
                ++chunk_done;
                ++stats.chunk_rounds;
            } else {
                if (L != 0) return fail(SA_HIP_EINTERNAL, "truncated build cannot make progress");
                if (gb + rb > 64) return fail(SA_HIP_EINTERNAL, "composite key exceeds 64 bits");
                if (!have_isa) {
                    if ((rc = isa.ensure((size_t)n * 4))) return rc;
                    if ((rc = scatter_ranks(flags.as<u8>(), n32, nullptr, sa))) return rc;
                    have_isa = true;
                }
                hipLaunchKernelGGL(doubling_keys_kernel, dim3(stream_grid(M, 256)), dim3(256), 0, stream, isa.as<u32>(), n,
                                   aidx.as<u32>(), gid.as<u32>(), M, h, rb, rkeys0.as<u64>());
                begin_bit = 0;
                h_next = 2 * h;
                ++stats.doubling_rounds;
            }
            // sort the active records; aidx is the value array (read in place by the tile-local sort)
            const int end_bit = use_chunk ? 64 : (gb + rb);
            if ((rc = round_sort(rkeys0.as<u64>(), aidx.as<u32>(), ridx0.as<u32>(), rkeys1.as<u64>(), ridx1.as<u32>(), M, G, begin_bit, end_bit,
                                 use_chunk ? (gb ? 64 - gb : 64) : rb, &kres, &vres))) return rc;
            // write back, new heads, counts
            if ((rc = flags_and_counts(kres, M, lf.as<u8>(), apos_cur, vres, tot))) return rc;
            if ((rc = check_device_status())) return rc;
            if (have_isa) {
                if ((rc = scatter_ranks(lf.as<u8>(), M, apos_cur, vres))) return rc;
            }
            if (debug_rounds) {
                u64 dups_sa = 0, dups_act = 0;
                if ((rc = dbg.ensure((size_t)n * 4 + 16))) return rc;
                u64* dd = reinterpret_cast<u64*>(small.as<u8>() + 3072);
                SA_HIP_CHECK(hipMemsetAsync(dbg.p, 0, (size_t)n * 4, stream));
                SA_HIP_CHECK(hipMemsetAsync(dd, 0, 49 * 8, stream));
                hipLaunchKernelGGL(dbg_dup_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, stream, (const u32*)sa, n, dbg.as<u32>(), dd);
                SA_HIP_CHECK(hipMemcpyAsync(&dups_sa, dd, 8, hipMemcpyDeviceToHost, stream));
                SA_HIP_CHECK(hipStreamSynchronize(stream));
                SA_HIP_CHECK(hipMemsetAsync(dbg.p, 0, (size_t)n * 4, stream));
                SA_HIP_CHECK(hipMemsetAsync(dd, 0, 49 * 8, stream));
                hipLaunchKernelGGL(dbg_dup_kernel, dim3(stream_grid(M, 1024)), dim3(256), 0, stream, (const u32*)vres, (u64)M, dbg.as<u32>(), dd);
                SA_HIP_CHECK(hipMemcpyAsync(&dups_act, dd, 8, hipMemcpyDeviceToHost, stream));
                SA_HIP_CHECK(hipStreamSynchronize(stream));
                fprintf(stderr, "[sa_hip] round %u %s h=%llu M=%u G=%u -> M'=%u G'=%u passes=%d dupSA=%llu dupActive=%llu\n", stats.rounds,
                        use_chunk ? "chunk" : "dbl", (unsigned long long)h, M, G, tot[0], tot[1], (end_bit - begin_bit + 7) / 8,
                        (unsigned long long)dups_sa, (unsigned long long)dups_act);
            }
            stats.active_total += M;
            ++stats.rounds;
            M_prev = M;
            h = h_next;
            const u32 M_next = tot[0];
            if (M_next && (L == 0 || h < L)) {
                launch_compact(lf.as<u8>(), M, M_next, apos_cur, vres, apos_nxt, aidx.as<u32>(), gid.as<u32>());
                swap_lists();
            }
            M = M_next;
            G = tot[1];
            if (stats.rounds > (debug_rounds ? 40u : 200u)) return fail(SA_HIP_EINTERNAL, "refinement did not converge");
        }
        stats.final_depth = (u32)(h > 0xFFFFFFFFull ? 0xFFFFFFFFull : h);
        if (sa64_out) {
            if (!stats.widen_fused)
                hipLaunchKernelGGL(widen_kernel, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, stream, (const u32*)sa, n, sa64_out);
            else if (M0 && !patched_by_tiny)
                hipLaunchKernelGGL(widen_patch_kernel, dim3(stream_grid(M0, 256)), dim3(256), 0, stream, (const u32*)apos0.as<u32>(), M0,
                                   (const u32*)sa, sa64_out);
        }
        if ((rc = build_directory())) return rc;
        return finish_stats();
    }

    int finish_stats() {
        SA_HIP_CHECK(hipEventRecord(ev_end, stream));
        SA_HIP_CHECK(hipEventSynchronize(ev_end));
        float ms = 0.f;
        SA_HIP_CHECK(hipEventElapsedTime(&ms, ev_begin, ev_end));
        int rc = radix.timer.flush();
        if (rc) return rc;
        stats.total_ms = ms;
        stats.radix_ms = radix.timer.total_ms;
        stats.radix_passes = (u32)radix.passes;
        stats.radix_records = radix.pass_records;
        stats.radix_bytes = radix.pass_bytes;
        for (int k = 0; k < PASS_KINDS; ++k) {
            stats.pass_ms[k] = radix.timer.kind_ms[k];
            stats.pass_bytes[k] = radix.timer.kind_bytes[k];
            stats.pass_launches[k] = (u32)radix.timer.kind_launches[k];
        }
        return 0;
    }
};

}  // namespace sa
