// group_finish.hpp -- tied groups that fit a tile are refined TO THE END by one workgroup, entirely in LDS.
//
// After the initial sort the active list holds the suffixes whose first h characters are shared with another suffix,
// in SA order, group by group.  A refinement round of sa_build.hpp appends the next characters to every active record
// and costs five full passes over the list (key generation, sort, flags + write-back, scan, compaction) plus a host
// round trip -- per round, for every record that is still tied.  On word / name / log-like text most groups are
// small (a few to a few hundred suffixes) and come apart within the next 10-30 characters, so here ONE launch does
// all of their rounds: the list is cut into tiles of whole groups exactly as for the tile-local round sort
// (round_sort.hpp: LocTile), and a workgroup keeps its tile's records in LDS and loops
//     fetch the next kc characters of every record that is still tied (one unaligned 8-byte text load)
//     stable LSD radix sort by (group, characters)  -- ballot-match ranking, the records stay in registers between passes
//     neighbours that differ start new groups; singletons are final and leave the working set (in-LDS compaction)
// until nothing is tied any more.  Only groups larger than a tile (and what an adversarial text leaves after
// `max_rounds` rounds or two rounds without any split) stay on the global path; those are left EXACTLY as they were
// (all writes of a tile are deferred to its end and skipped for such a group), so the global rounds keep one uniform
// depth h for everything that is still active.
//
// Replaces, by function only, the recursion of the reference's truncated builder into small buckets
// (engine.c:696-812: MSD pass per depth, insertion sort below 32) and what libsais' induced sorting does for the
// same suffixes (libsais.c:3813-3820, 5814-5836); order: unsigned bytes, a suffix that ends sorts first,
// truncated builds stop at depth L with ties in text order (every pass is stable and the list starts in text order
// within a group).
#pragma once
#include "round_sort.hpp"

namespace sa {

// Tile capacity and workgroup size (LDS per tile = 18 bytes per record of capacity + the wave histograms).  Measured on
// words / names 1e8 (tools/gpu_fin_sweep2.py): 4096 x 8 per thread (512 threads, two workgroups per CU) and 2048 x 8 (256
// threads, four per CU) run alike -- 11.5 vs 11.6 ms, 7.8 vs 7.9 ms: the kernel is bound by its random text fetches (about
// 1.25 per record it looks at, each a 64-byte sector), not by occupancy -- and the larger tile keeps more groups off the
// global path; 4 records per thread (twice the barriers per record): +20 %, 16 per thread: +15 %.
#ifndef SA_FIN_CAP
#define SA_FIN_CAP 4096
#endif
#ifndef SA_FIN_ITEMS
#define SA_FIN_ITEMS 8
#endif
constexpr u32 FIN_CAP = SA_FIN_CAP;                     // records per tile (power of two)
constexpr int FIN_ITEMS = SA_FIN_ITEMS;                 // records per thread
constexpr int FIN_BLOCK = (int)(FIN_CAP / FIN_ITEMS);
constexpr u32 FIN_TILE = FIN_CAP - FIN_CAP / 8;         // nominal tile length (round_sort.hpp: tiles of whole groups)
constexpr int fin_log2(u32 v) { return v <= 1 ? 0 : 1 + fin_log2(v >> 1); }
constexpr int FIN_POS_BITS = fin_log2(FIN_CAP);         // a record's index in the tile
constexpr u32 FIN_MAX_ROUNDS = 24;
static_assert((1u << FIN_POS_BITS) == FIN_CAP && FIN_ITEMS % 2 == 0 && FIN_BLOCK % WAVE == 0 && FIN_BLOCK >= RADIX, "tile geometry");
static_assert((FIN_BLOCK / WAVE) * FIN_ITEMS <= 64, "at most one lane per row in the compaction scan");

struct FinArgs {
    const u8* text;
    u64 n;
    int b;                 // bits per character code
    const u32* aidx;       // active list: suffix index
    const u32* gid;        //              dense group id (ascending)
    const u32* apos;       //              SA slot
    const LocTile* tiles;  // records [begin, local_end) of tile t are whole groups, at most FIN_CAP of them
    u32 h0;                // characters every group is known to share
    u32 L;                 // truncated build: order by the first L characters only (0 = full)
    u32 max_rounds;
    u32 count_max;         // a round whose largest group has at most this many records is ordered by counting
    int radix_chars;       // characters per round while larger groups exist (radix rounds; 0 = as many as fit)
    u32* sa;               // out: suffix array slots of resolved groups
    u8* gflags;            // out: bit0 = group head, per SA slot
    u8* done;              // out: per list element 1 = final (its group was resolved and written), 0 = untouched
    u32* res_idx;          // scratch [M]: final suffix per list position
    u8* res_fin;           // scratch [M]: 0 not final, 1 final, 3 final + head of a (sub)group
    const u64* w0;         // group_finish2_kernel: the 8 text bytes at suffix + h0 of every list position inside a tile (fin_prefetch_kernel)
    unsigned long long* totals;   // [0] records in tiles, [1] records resolved; debug: [2] rounds, [3] radix rounds,
                                  // [4] record slots of the radix rounds, [5] of the counting rounds, [6] tiles, [7] active records over all rounds
    int debug;
};

__device__ __forceinline__ int fin_bits_for(u32 count) {   // smallest g with 2^g >= count
    return count <= 1 ? 0 : 32 - __clz((int)(count - 1));
}

// wave_rank (radix_sort.hpp) over the first `rows` of the ITEMS wave-striped rows of a lane (rows even, uniform): a round
// costs what is still tied, not the tile
template <int ITEMS>
__device__ __forceinline__ void fin_wave_rank(const u64 (&key)[ITEMS], int shift, u32 mask, u32* wh, u32 (&rd)[ITEMS], int rows) {
#pragma unroll
    for (int j = 0; j < ITEMS; j += 2) {
        if (j < rows) {
            u32 d[2], lo[2], hi[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                d[i] = (u32)(key[j + i] >> shift) & mask;
                lo[i] = ~0u; hi[i] = ~0u;
            }
#pragma unroll
            for (int b = 0; b < RADIX_BITS; ++b) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    u32 e = (u32)__builtin_amdgcn_sbfe((int)d[i], b, 1);
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
                if (below == 0) wh[d[i]] = prior + total;
                __builtin_amdgcn_wave_barrier();
                rd[j + i] = (prior + below) | (d[i] << 16);
            }
        }
    }
}

// groups of at most FIN_COUNT_MAX records are ordered by counting (rank = members with a smaller key) instead of radix
// passes: for the typical handful of members that is a few dozen instructions per record against ~280 for seven passes
constexpr u32 FIN_COUNT_MAX = 96;   // 0 / 48 / 96 / 160: words 1e8 11.5 / 11.0 / 10.8 / 10.9 ms
constexpr int FIN_RADIX_CHARS = 0;

__global__ __launch_bounds__(FIN_BLOCK, FIN_BLOCK == 512 ? 4 : 2) void group_finish_kernel(FinArgs a, CodeMap map) {
    constexpr int WAVES = FIN_BLOCK / WAVE;
    constexpr int ITEMS = FIN_ITEMS;
    constexpr u32 CAP = FIN_CAP;
    __shared__ u64 s_key[CAP];
    __shared__ u32 s_idx[CAP];          // by compact index: suffix
    __shared__ u16 s_lgid[CAP];         // by compact index: dense id of the record's group among the active groups
    __shared__ u16 s_gq[CAP / 2 + 2];   // by group: compact index of its first record; [G] = A  (an active group has >= 2 members)
    __shared__ u16 s_gpos[CAP / 2 + 2]; // by group: position in the tile of its first record (members are contiguous there too)
    __shared__ u32 s_whist[WAVES * RADIX];
    __shared__ u32 s_wsum[RADIX / WAVE];
    __shared__ u32 s_rowa[64], s_rowh[64];
    __shared__ u32 s_fail[CAP / 32];    // original groups that were not resolved
    __shared__ u16 s_map[256];          // codes reach 256 when every byte value occurs (b = 9)
    __shared__ u32 s_A, s_G, s_maxg;

    const LocTile lt = a.tiles[blockIdx.x];
    const u32 cnt = lt.local_end - lt.begin;   // <= CAP
    if (cnt == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 begin = lt.begin;
    const u32 gid0 = a.gid[begin];

    if (tid < 256) s_map[tid] = map.code[tid];
    for (u32 i = tid; i < CAP / 32; i += FIN_BLOCK) s_fail[i] = 0;
    const u32 G0 = a.gid[begin + cnt - 1] - gid0 + 1;
    {   // the tile: every load of a thread is requested before the first is used (clamped indices; cnt <= CAP = ITEMS * FIN_BLOCK);
        // group starts are found from the LDS copy afterwards (a third load per record made the kernel spill)
        u32 gv[ITEMS], iv[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            u32 p = (u32)tid + (u32)it * FIN_BLOCK;
            asm("" : "+v"(p));   // keeps these addresses from being shared with (and kept alive until) the write-out at the end
            const u32 pc = p < cnt ? p : cnt - 1;
            gv[it] = a.gid[begin + pc];
            iv[it] = a.aidx[begin + pc];
        }
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            u32 p = (u32)tid + (u32)it * FIN_BLOCK;
            asm("" : "+v"(p));
            if (p < cnt) {
                s_idx[p] = iv[it];
                s_lgid[p] = (u16)(gv[it] - gid0);
                a.res_fin[begin + p] = 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            u32 p = (u32)tid + (u32)it * FIN_BLOCK;
            asm("" : "+v"(p));
            if (p < cnt) {
                const u32 g = s_lgid[p];
                if (p == 0 || (u32)s_lgid[p - 1] != g) { s_gq[g] = (u16)p; s_gpos[g] = (u16)p; }
            }
        }
    }
    if (tid == 0) { s_A = cnt; s_G = G0; s_gq[G0] = (u16)cnt; }
    __syncthreads();
    if (a.debug) {   // totals[16 + c]: records in groups of [2^c, 2^(c+1)) members when the finisher takes them over
        if (tid < 16) s_whist[tid] = 0;
        __syncthreads();
        for (u32 g = tid; g < G0; g += FIN_BLOCK) {
            const u32 sz = (u32)s_gq[g + 1] - (u32)s_gq[g];
            atomicAdd(&s_whist[31 - __clz((int)sz)], sz);
        }
        sync_lds();
        if (tid < 16 && s_whist[tid]) atomicAdd(&a.totals[16 + tid], (unsigned long long)s_whist[tid]);
        __syncthreads();
    }

    u32 h = a.h0, rounds = 0, stall = 0;
    u32 dbg_radix = 0, dbg_slots_r = 0, dbg_slots_c = 0, dbg_act = 0;
    // SA_HIP_DEBUG_ROUNDS=1: cycles of thread 0 per phase (fetch + keys / counting sort / radix sort / regroup), totals[8..11]
    long long t_fetch = 0, t_count = 0, t_radix = 0, t_regroup = 0, t_mark = a.debug ? clock64() : 0;
    u32* wh = s_whist + wave * RADIX;
    const u32 inv_b = 65536u / (u32)a.b + 1u;   // x / b == (x * inv_b) >> 16 for x <= 64, b <= 9
    while (true) {
        // the round's control values as SCALARS: read from LDS they sit in vector registers, and everything derived from them
        // (rows per lane, characters per round, passes) turned every `if (j < R)` / `if (c < kc)` below into exec-mask branches
        const u32 A = (u32)__builtin_amdgcn_readfirstlane((int)s_A), G = (u32)__builtin_amdgcn_readfirstlane((int)s_G);
        if (A == 0) break;
        if (rounds == a.max_rounds || stall >= 2) {
            // what is still tied goes back to the global path: its ORIGINAL group is left as it was
            for (u32 g = tid; g < G; g += FIN_BLOCK) {
                const u32 og = a.gid[begin + s_gpos[g]] - gid0;
                atomicOr(&s_fail[og >> 5], 1u << (og & 31));
            }
            break;
        }
        // rows per lane this round: the working set is wave-striped over ALL waves (wave w: records [w*64*R, (w+1)*64*R))
        int R = (int)((A + (u32)(WAVE * WAVES) - 1) / (u32)(WAVE * WAVES));
        R = (R + 1) & ~1;
        const u32 woff = (u32)wave * (u32)(WAVE * R) + lane;
        if (tid == 0) s_maxg = 0;
        if (tid < 64) { s_rowa[tid] = 0; s_rowh[tid] = 0; }
        __syncthreads();
        {   // largest active group (decides how this round is sorted)
            u32 mx = 0;
            for (u32 g = tid; g < G; g += FIN_BLOCK) { const u32 sz = (u32)s_gq[g + 1] - (u32)s_gq[g]; mx = sz > mx ? sz : mx; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const u32 t = __shfl_down(mx, o); mx = t > mx ? t : mx; }
            if (lane == 0 && mx) atomicMax(&s_maxg, mx);
        }
        sync_lds();   // s_maxg complete (LDS atomics above)
        const bool counting = (u32)__builtin_amdgcn_readfirstlane((int)s_maxg) <= a.count_max;
        if (counting) dbg_slots_c += (u32)(WAVE * WAVES * R); else { dbg_slots_r += (u32)(WAVE * WAVES * R); ++dbg_radix; }
        dbg_act += A;
        const int gbits = fin_bits_for(G);
        int kc = (int)(((u32)(64 - FIN_POS_BITS - gbits) * inv_b) >> 16);
        if (kc > 8) kc = 8;
        if (!counting && a.radix_chars > 0 && kc > a.radix_chars) kc = a.radix_chars;   // cheap rounds until the large groups are apart
        if (a.L && (u32)kc > a.L - h) kc = (int)(a.L - h);   // h < L while anything is active
        const int cbits = kc * a.b;
        const bool last_trunc = a.L && (h + (u32)kc >= a.L);

        // 1. keys: (group << cbits | next kc characters) above the record's compact index
        u64 key[ITEMS];
        u64 wtext[ITEMS];   // all text fetches of a lane in flight together (inactive rows read the text's first word: cached)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            wtext[j] = 0;
            if (j < R) {   // uniform
                const u32 q = woff + j * WAVE;
                const u64 start = (q < A) ? (u64)s_idx[q] + h : 0ull;
                __builtin_memcpy(&wtext[j], a.text + start, 8);      // the text is zero padded: in bounds
            }
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            key[j] = ~0ull;   // padding sorts last in every pass and stays behind the records (stable)
            if (j < R) {
                const u32 q = woff + j * WAVE;
                if (q < A) {
                    const u64 start = (u64)s_idx[q] + h;             // <= n: the group shares h real characters
                    const u64 avail = a.n - start;
                    const u64 w = wtext[j];
                    u64 chars = 0;
                    if (a.b <= 8) {   // uniform
                        // all eight codes looked up at once (eight independent LDS reads), packed as two 32-bit halves, the
                        // first kc kept by one shift; codes past the end of the text are cleared by a mask (rare)
                        const u32 w0 = (u32)w, w1 = (u32)(w >> 32);
                        const u32 c0 = s_map[w0 & 255u], c1 = s_map[(w0 >> 8) & 255u], c2 = s_map[(w0 >> 16) & 255u], c3 = s_map[w0 >> 24];
                        const u32 c4 = s_map[w1 & 255u], c5 = s_map[(w1 >> 8) & 255u], c6 = s_map[(w1 >> 16) & 255u], c7 = s_map[w1 >> 24];
                        const u32 hi4 = (((((c0 << a.b) | c1) << a.b) | c2) << a.b) | c3;
                        const u32 lo4 = (((((c4 << a.b) | c5) << a.b) | c6) << a.b) | c7;
                        u64 all = ((u64)hi4 << (4 * a.b)) | lo4;          // character 0 on top, 8 * b <= 64 bits
                        if (avail < 8) all = avail ? (all & (~0ull << (a.b * (8 - (int)avail)))) : 0ull;
                        chars = all >> (a.b * (8 - kc));                   // 1 <= kc <= 8
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            if (c < kc) {
                                const u32 byte = (u32)(w >> (8 * c)) & 255u;
                                const u32 code = ((u64)c < avail) ? (u32)s_map[byte] : 0u;
                                chars = (chars << a.b) | code;
                            }
                        }
                    }
                    key[j] = ((((u64)s_lgid[q] << cbits) | chars) << FIN_POS_BITS) | q;
                }
            }
        }
        if (a.debug) { const long long t = clock64(); t_fetch += t - t_mark; t_mark = t; }
        if (counting) {
            // 2a. every group is small: a record's place inside its group = the members with a smaller key (the keys are
            //     distinct: they end in the compact index, which also makes the order stable)
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                if (j < R) { const u32 q = woff + j * WAVE; if (q < A) s_key[q] = key[j]; }
            }
            __syncthreads();
            u32 dst[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                dst[j] = 0;
                if (j < R) {
                    const u32 q = woff + j * WAVE;
                    if (q < A) {
                        const u32 g = s_lgid[q];
                        const u32 q0 = s_gq[g], q1 = s_gq[g + 1];
                        // four members per step, their loads in flight together (one member per step waits a whole LDS round trip per
                        // member; the wave runs as long as its largest group); reads past the group are masked out
                        u32 rank = 0;
                        const u64 kj = key[j];
                        for (u32 m = q0; m < q1; m += 4) {
                            const u64 k0 = s_key[m], k1 = s_key[(m + 1) & (CAP - 1u)], k2 = s_key[(m + 2) & (CAP - 1u)],
                                      k3 = s_key[(m + 3) & (CAP - 1u)];
                            rank += (k0 < kj ? 1u : 0u) + ((m + 1 < q1 && k1 < kj) ? 1u : 0u) + ((m + 2 < q1 && k2 < kj) ? 1u : 0u) +
                                    ((m + 3 < q1 && k3 < kj) ? 1u : 0u);
                        }
                        dst[j] = q0 + rank;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                if (j < R) { const u32 q = woff + j * WAVE; if (q < A) s_key[dst[j]] = key[j]; }
            }
            __syncthreads();
        } else {
            // 2b. stable LSD radix sort of the working set by bits [FIN_POS_BITS, FIN_POS_BITS + cbits + gbits)
            const int top = FIN_POS_BITS + cbits + gbits;
            const int passes = (top - FIN_POS_BITS + RADIX_BITS - 1) / RADIX_BITS;
            for (int pass = 0; pass < passes; ++pass) {
                const int shift = FIN_POS_BITS + RADIX_BITS * pass;
                const int bits = (top - shift) < RADIX_BITS ? (top - shift) : RADIX_BITS;
                const u32 mask = (1u << bits) - 1u;
                for (int i = tid; i < WAVES * RADIX; i += FIN_BLOCK) s_whist[i] = 0;
                __syncthreads();
                u32 rd[ITEMS];
                fin_wave_rank<ITEMS>(key, shift, mask, wh, rd, R);
                __syncthreads();
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
#pragma unroll
                for (int j = 0; j < ITEMS; ++j)
                    if (j < R) s_key[wh[rd[j] >> 16] + (rd[j] & 0xFFFFu)] = key[j];
                __syncthreads();
                if (pass + 1 < passes) {
                    // (no barrier after these reads: the next write to s_key is the next pass's placement, three barriers on;
                    //  the counters it zeroes first were last read before the barrier above)
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j)
                        if (j < R) key[j] = s_key[woff + j * WAVE];
                }
            }
        }
        if (a.debug) { const long long t = clock64(); if (counting) t_count += t - t_mark; else t_radix += t - t_mark; t_mark = t; }
        // 3. new groups, finals, compaction of what is still tied (row = the 64 records of one (wave, j))
        u32 r_idx[ITEMS], r_pos[ITEMS];
        u64 m_act[ITEMS], m_head[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            m_act[j] = 0; m_head[j] = 0; r_idx[j] = 0; r_pos[j] = 0;
            if (j < R) {
                const u32 q = woff + j * WAVE;
                const bool valid = q < A;
                u64 k = 0, kp = 0, kn = 0;
                if (valid) {
                    k = s_key[q] >> FIN_POS_BITS;
                    kp = (q > 0) ? (s_key[q - 1] >> FIN_POS_BITS) : ~0ull;
                    kn = (q + 1 < A) ? (s_key[q + 1] >> FIN_POS_BITS) : ~0ull;
                }
                const bool head = valid && (q == 0 || k != kp);
                const bool single = head && (q + 1 >= A || kn != k);
                const bool fin = valid && (single || last_trunc);
                if (valid) {
                    r_idx[j] = s_idx[(u32)s_key[q] & (CAP - 1u)];   // the record that now stands at q
                    const u32 g = s_lgid[q];                        // q's group is the one it had before: sorting stays inside groups
                    r_pos[j] = (u32)s_gpos[g] + (q - (u32)s_gq[g]);
                }
                if (fin) {
                    a.res_idx[begin + r_pos[j]] = r_idx[j];
                    a.res_fin[begin + r_pos[j]] = head ? (u8)3 : (u8)1;
                }
                const bool act = valid && !fin;
                m_act[j] = __ballot(act);
                m_head[j] = __ballot(act && head);
                if (lane == 0) { s_rowa[wave * ITEMS + j] = (u32)__popcll(m_act[j]); s_rowh[wave * ITEMS + j] = (u32)__popcll(m_head[j]); }
            }
        }
        __syncthreads();   // every read of s_key / s_idx / s_lgid / s_gq / s_gpos of this round is done
        u32 ia = s_rowa[lane], ih = s_rowh[lane];   // every wave scans the 64 row counts
        const u32 ca = ia, ch = ih;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 ta = __shfl_up(ia, o), th = __shfl_up(ih, o);
            if (lane >= o) { ia += ta; ih += th; }
        }
        const u32 tot_a = (u32)__builtin_amdgcn_readlane((int)ia, 63), tot_h = (u32)__builtin_amdgcn_readlane((int)ih, 63);
        const u32 ea = ia - ca, eh = ih - ch;
        const u64 lt_mask = lanemask_lt();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (j < R) {
                const u32 row = (u32)wave * ITEMS + j;
                const u32 base_a = __shfl(ea, (int)row), base_h = __shfl(eh, (int)row);
                if ((m_act[j] >> lane) & 1ull) {
                    const u32 na = base_a + (u32)__popcll(m_act[j] & lt_mask);
                    const u32 ng = base_h + (u32)__popcll(m_head[j] & (lt_mask | (1ull << lane))) - 1u;
                    s_idx[na] = r_idx[j];
                    s_lgid[na] = (u16)ng;
                    if ((m_head[j] >> lane) & 1ull) { s_gq[ng] = (u16)na; s_gpos[ng] = (u16)r_pos[j]; }
                }
            }
        }
        stall = (tot_a < A || tot_h > G) ? 0u : stall + 1u;
        if (tid == 0) { s_A = tot_a; s_G = tot_h; s_gq[tot_h] = (u16)tot_a; }
        __syncthreads();
        h += (u32)kc;
        ++rounds;
        if (a.debug) { const long long t = clock64(); t_regroup += t - t_mark; t_mark = t; }
    }
    __syncthreads();   // s_fail complete; the finals of every lane have reached memory (workgroup scope)
    u32 resolved = 0;
    {
        constexpr int HALF = ITEMS / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32 ogv[HALF], slotv[HALF], idxv[HALF];
            u8 fv[HALF];
#pragma unroll
            // (a loop with a run-time trip count and loads under `if` serialises the memory latencies of its iterations: the
            //  loads of four records of a thread are issued first, with clamped indices, then used.  Round 3, SA_HIP_DEBUG_ROUNDS=1:
            //  the text fetches and this write-out were 55 % of a tile's time.  Doing the same to the tile load at the top takes the
            //  kernel from 123 to 175 VGPRs -- one workgroup per CU -- so that loop stays as it is.)
            for (int it = 0; it < HALF; ++it) {
                const u32 p = (u32)tid + (u32)(half * HALF + it) * FIN_BLOCK;
                const u32 pc = p < cnt ? p : cnt - 1;
                ogv[it] = a.gid[begin + pc] - gid0;
                fv[it] = a.res_fin[begin + pc];
                slotv[it] = a.apos[begin + pc];
                idxv[it] = a.res_idx[begin + pc];
            }
#pragma unroll
            for (int it = 0; it < HALF; ++it) {
                const u32 p = (u32)tid + (u32)(half * HALF + it) * FIN_BLOCK;
                if (p < cnt) {
                    const u32 og = ogv[it];
                    const u8 f = fv[it];
                    const bool ok = !((s_fail[og >> 5] >> (og & 31)) & 1u) && f != 0;
                    if (ok) {
                        a.sa[slotv[it]] = idxv[it];
                        if (f & 2) a.gflags[slotv[it]] = 1;
                        ++resolved;
                    }
                    a.done[begin + p] = ok ? (u8)1 : (u8)0;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) resolved += __shfl_down(resolved, o);
    if (lane == 0 && resolved) atomicAdd(&a.totals[1], (unsigned long long)resolved);
    if (tid == 0) {
        atomicAdd(&a.totals[0], (unsigned long long)cnt);
        if (a.debug) {
            atomicAdd(&a.totals[2], (unsigned long long)rounds); atomicAdd(&a.totals[3], (unsigned long long)dbg_radix);
            atomicAdd(&a.totals[4], (unsigned long long)dbg_slots_r); atomicAdd(&a.totals[5], (unsigned long long)dbg_slots_c);
            atomicAdd(&a.totals[6], 1ull); atomicAdd(&a.totals[7], (unsigned long long)dbg_act);
            atomicAdd(&a.totals[8], (unsigned long long)t_fetch); atomicAdd(&a.totals[9], (unsigned long long)t_count);
            atomicAdd(&a.totals[10], (unsigned long long)t_radix); atomicAdd(&a.totals[11], (unsigned long long)t_regroup);
            atomicAdd(&a.totals[12], (unsigned long long)(clock64() - t_mark));
        }
    }
}

// ---- what a finisher run leaves behind, without a pass over the list (round 4) ------------------------------------------------
// When no tile gave up on a group, the records a run leaves are exactly the groups its plan left out: records [local_end, end)
// of every tile, one group each (round_sort.hpp: LocTile).  The next lists are then a copy of those ranges -- work in proportion
// to what is LEFT (config 5: 56 M of 530 M records) -- instead of done flags for all M records, a flags pass, a scan and a
// compaction over them.  left_scan_kernel: exclusive prefixes of the ranges' sizes (into LocTile::big_off) and of the non-empty
// ranges (the next dense group ids), totals {records, groups}; left_copy_kernel: the ranges to the next lists.
__global__ __launch_bounds__(1024) void left_scan_kernel(LocTile* __restrict__ tiles, u32 ntiles, u32* __restrict__ next_gid, u32* __restrict__ totals) {
    __shared__ u32 s_a[16], s_b[16];
    __shared__ u32 s_ca, s_cb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_ca = 0; s_cb = 0; }
    __syncthreads();
    for (u32 base = 0; base < ntiles; base += 1024) {
        const u32 t = base + threadIdx.x;
        const u32 va = (t < ntiles) ? tiles[t].end - tiles[t].local_end : 0u;
        const u32 vb = va ? 1u : 0u;
        u32 ia = va, ib = vb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 xa = __shfl_up(ia, o), xb = __shfl_up(ib, o);
            if (lane >= o) { ia += xa; ib += xb; }
        }
        if (lane == 63) { s_a[wave] = ia; s_b[wave] = ib; }
        __syncthreads();
        u32 oa = s_ca, ob = s_cb;
        for (int w = 0; w < wave; ++w) { oa += s_a[w]; ob += s_b[w]; }
        if (t < ntiles) { tiles[t].big_off = oa + ia - va; next_gid[t] = ob + ib - vb; }
        __syncthreads();
        if (threadIdx.x == 1023) { s_ca = oa + ia; s_cb = ob + ib; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = s_ca; totals[1] = s_cb; }
}
__global__ __launch_bounds__(256) void left_copy_kernel(const LocTile* __restrict__ tiles, u32 ntiles, const u32* __restrict__ next_gid,
                                                        const u32* __restrict__ apos, const u32* __restrict__ aidx, u32* __restrict__ out_apos,
                                                        u32* __restrict__ out_aidx, u32* __restrict__ out_gid) {
    for (u32 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const LocTile lt = tiles[t];
        const u32 cnt = lt.end - lt.local_end;
        if (cnt == 0) continue;
        const u32 per = (cnt + gridDim.y - 1) / gridDim.y;
        const u32 lo = blockIdx.y * per;
        const u32 hi = (lo + per < cnt) ? lo + per : cnt;
        const u32 g = next_gid[t];
        for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const u64 r = (u64)lt.local_end + i, l = (u64)lt.big_off + i;
            out_apos[l] = apos[r];
            out_aidx[l] = aidx[r];
            out_gid[l] = g;
        }
    }
}

// ---- round 4: the finisher restructured (SA_HIP_FIN_V2=1; NOT the default) ---------------------------------------------------------
// MEASURED (profiles/r04_finisher_v2_ab.log): bit-identical to the kernel above on every test and no faster -- names 2e8 13.5 vs
// 13.5 ms, words 1e8 10.1 vs 10.1, the config-5 column 62.6-62.9 vs 61.4 ms.  The counters say why: both kernels issue the same
// 4 700 vector instructions per wave and tile (the vector ALUs are busy 52 % of the time, the rest is latency two workgroups per
// CU cannot hide); what the splits save in radix passes the additional cheap rounds spend again (3.1 instead of 2.5 rounds per
// tile: a split group needs one more round before counting takes over).  Kept behind the switch, with its parity test.
// What the phase timers of the kernel above said (names, config 5): a round whose tile holds ONE group of more than
// `count_max` members sends the whole working set through seven radix passes (57 K of a tile's 167 K cycles for 0.9 such
// rounds per tile, against 10 K for the 2.7 counting rounds), and the deferred write-out re-reads four arrays per record
// (27 K).  Here:
//   * every group has its OWN depth (s_gh: characters beyond h0 its members share), so groups of one tile need not advance
//     in step;
//   * a group of more than `count_max` members is split apart from the rest, by ONE WAVE and without a workgroup barrier:
//     the wave finds the common prefix of the group's fetched characters (an OR of differences), ranks the members by the
//     FIN_SPLIT bits that follow it (stable ballot-match ranking, per-wave counters) and leaves every member's place in
//     the low bits of its key; the group advances by the whole characters those bits cover (a run of characters every
//     member shares -- "INTERNATION|AL " -- costs nothing), its parts are ordinary groups from the next round on;
//   * everything else is ordered by counting as before (a record's place = the members of its group with a smaller key);
//   * a final record goes straight to its SA slot (one load of the slot, three stores) instead of through per-position
//     scratch arrays that the end of the tile read back; the rare tile that gives up on a group (24 rounds, or two without
//     a split: long repeats) RESTORES that group's slots from the list, which still holds what they were.
// A key is  characters (FIN_FIELD bits, left aligned, masked to the group's kc characters) | compact index (12) | place (12).
#ifndef SA_FIN_SPLIT_BITS
#define SA_FIN_SPLIT_BITS 8
#endif
constexpr int FIN_SPLIT = SA_FIN_SPLIT_BITS;
constexpr int FIN_FIELD = 64 - 2 * FIN_POS_BITS;
constexpr u32 FIN_BIG_CAP = 256;          // groups of more than 16 members per tile: at most CAP / 17
static_assert(FIN_CAP / 17 < FIN_BIG_CAP && FIN_SPLIT >= 4 && FIN_SPLIT <= 10, "big-group list / split digit");

// The first round's text fetches of every tile as a kernel of their own: one random 8-byte read per record and nothing else,
// thousands of them in flight per CU -- inside the finisher the same reads are issued by two workgroups per CU that then sit
// waiting for them (the finisher's time did not move when its sort or its write-out were restructured: it was waiting here).
// w0[p] = the 8 bytes at text + aidx[p] + h0 for the list positions p inside a tile (groups too large for a tile are skipped).
__global__ __launch_bounds__(256) void fin_prefetch_kernel(const u8* __restrict__ text, const u32* __restrict__ aidx, const LocTile* __restrict__ tiles,
                                                           u32 h0, u64* __restrict__ w0) {
    const LocTile lt = tiles[blockIdx.x];
    const u32 begin = lt.begin, cnt = lt.local_end - lt.begin;
    for (u32 p0 = 0; p0 < cnt; p0 += 4 * 256) {
        u32 iv[4];
        u64 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 p = p0 + u * 256 + threadIdx.x;
            iv[u] = aidx[begin + (p < cnt ? p : cnt - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_memcpy(&w[u], text + (u64)iv[u] + h0, 8);   // the text is zero padded: in bounds
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 p = p0 + u * 256 + threadIdx.x;
            if (p < cnt) w0[begin + p] = w[u];
        }
    }
}

__global__ __launch_bounds__(FIN_BLOCK, 4) void group_finish2_kernel(FinArgs a, CodeMap map) {
    constexpr int WAVES = FIN_BLOCK / WAVE;
    constexpr int ITEMS = FIN_ITEMS;
    constexpr u32 CAP = FIN_CAP;
    constexpr u32 PM = CAP - 1u;
    __shared__ u64 s_key[CAP];
    __shared__ u32 s_idx[CAP];            // by compact index: suffix
    __shared__ u16 s_lgid[CAP];           // by compact index: dense id of the record's group among the active groups
    __shared__ u32 s_gqp[CAP / 2 + 2];    // by group: compact index of its first record | its position in the tile << 16;  [G] = A
    __shared__ u8 s_gh[CAP / 2 + 2];      // by group: characters beyond h0 its members are known to share
    __shared__ u8 s_gsig[CAP / 2 + 2];    // by group: key bits ordered this round (0: all kc characters -- counting, or nothing to split)
    __shared__ u16 s_hist[WAVES << FIN_SPLIT];
    __shared__ u16 s_big[FIN_BIG_CAP];
    __shared__ u32 s_rowa[64], s_rowh[64];
    __shared__ u32 s_fail[CAP / 32];      // original groups that were not resolved
    __shared__ u16 s_map[256];
    __shared__ u32 s_A, s_G, s_nbig, s_anyfail;

    const LocTile lt = a.tiles[blockIdx.x];
    const u32 cnt = lt.local_end - lt.begin;   // <= CAP
    if (cnt == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 begin = lt.begin;
    const u32 gid0 = a.gid[begin];

    if (tid < 256) s_map[tid] = map.code[tid];
    for (u32 i = tid; i < CAP / 32; i += FIN_BLOCK) s_fail[i] = 0;
    const u32 G0 = a.gid[begin + cnt - 1] - gid0 + 1;
    for (u32 g = tid; g <= G0; g += FIN_BLOCK) { s_gh[g] = 0; s_gsig[g] = 0; }
    {
        u32 gv[ITEMS], iv[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            u32 p = (u32)tid + (u32)it * FIN_BLOCK;
            asm("" : "+v"(p));
            const u32 pc = p < cnt ? p : cnt - 1;
            gv[it] = a.gid[begin + pc];
            iv[it] = a.aidx[begin + pc];
        }
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            u32 p = (u32)tid + (u32)it * FIN_BLOCK;
            asm("" : "+v"(p));
            if (p < cnt) {
                s_idx[p] = iv[it];
                s_lgid[p] = (u16)(gv[it] - gid0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            u32 p = (u32)tid + (u32)it * FIN_BLOCK;
            asm("" : "+v"(p));
            if (p < cnt) {
                const u32 g = s_lgid[p];
                if (p == 0 || (u32)s_lgid[p - 1] != g) s_gqp[g] = p | (p << 16);
            }
        }
    }
    if (tid == 0) { s_A = cnt; s_G = G0; s_gqp[G0] = cnt; s_nbig = 0; s_anyfail = 0; }
    __syncthreads();
    if (a.debug) {   // totals[16 + c]: records in groups of [2^c, 2^(c+1)) members when the finisher takes them over
        u32* hh = reinterpret_cast<u32*>(s_hist);
        if (tid < 16) hh[tid] = 0;
        __syncthreads();
        for (u32 g = tid; g < G0; g += FIN_BLOCK) {
            const u32 sz = (s_gqp[g + 1] & 0xFFFFu) - (s_gqp[g] & 0xFFFFu);
            atomicAdd(&hh[31 - __clz((int)sz)], sz);
        }
        sync_lds();
        if (tid < 16 && hh[tid]) atomicAdd(&a.totals[16 + tid], (unsigned long long)hh[tid]);
        __syncthreads();
    }

    u32 rounds = 0, stall = 0, resolved = 0;
    u32 dbg_bigrounds = 0, dbg_bigrec = 0, dbg_slots = 0, dbg_act = 0;
    long long t_fetch = 0, t_sort = 0, t_perm = 0, t_regroup = 0, t_mark = a.debug ? clock64() : 0;
    const u32 ub = (u32)a.b;
    const u32 inv_b = 65536u / ub + 1u;                 // x / b == (x * inv_b) >> 16 for x <= 64, b <= 9
    u32 kc0 = ((u32)FIN_FIELD * inv_b) >> 16;           // whole characters a key holds (8 bytes are fetched)
    if (kc0 > 8) kc0 = 8;
    const u32 cm = a.count_max < 16u ? 16u : a.count_max;
    u16* wh = s_hist + ((u32)wave << FIN_SPLIT);
    while (true) {
        const u32 A = (u32)__builtin_amdgcn_readfirstlane((int)s_A), G = (u32)__builtin_amdgcn_readfirstlane((int)s_G);
        if (A == 0) break;
        if (rounds == a.max_rounds || stall >= 2) {
            // what is still tied goes back to the global path: its ORIGINAL group is restored below
            for (u32 g = tid; g < G; g += FIN_BLOCK) {
                const u32 og = a.gid[begin + (s_gqp[g] >> 16)] - gid0;
                atomicOr(&s_fail[og >> 5], 1u << (og & 31));
            }
            if (tid == 0) s_anyfail = 1;
            break;
        }
        const int R = (int)((A + (u32)(WAVE * WAVES) - 1) / (u32)(WAVE * WAVES));   // (no pairing of rows here: odd counts are fine)
        const u32 woff = (u32)wave * (u32)(WAVE * R) + lane;
        if (tid < 64) { s_rowa[tid] = 0; s_rowh[tid] = 0; }
        // the groups that are split apart this round
        for (u32 g = tid; g < G; g += FIN_BLOCK) {
            const u32 sz = (s_gqp[g + 1] & 0xFFFFu) - (s_gqp[g] & 0xFFFFu);
            if (sz > cm) { const u32 slot = atomicAdd(&s_nbig, 1u); s_big[slot] = (u16)g; }
        }
        dbg_slots += (u32)(WAVE * WAVES * R);
        dbg_act += A;

        // 1. keys: the next kc characters of every record, left aligned, above its compact index
        u64 key[ITEMS];
        u64 wtext[ITEMS];
        u32 hsum[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            wtext[j] = 0; hsum[j] = 0;
            if (j < R) {   // uniform
                const u32 q = woff + j * WAVE;
                if (rounds == 0 && a.w0) {   // (uniform) compact index = position in the tile, depth h0: fetched ahead by fin_prefetch_kernel
                    hsum[j] = a.h0;
                    wtext[j] = a.w0[begin + (q < A ? q : 0u)];
                } else {
                    u64 start = 0;
                    if (q < A) {
                        hsum[j] = a.h0 + (u32)s_gh[s_lgid[q]];
                        start = (u64)s_idx[q] + hsum[j];
                    }
                    __builtin_memcpy(&wtext[j], a.text + start, 8);      // the text is zero padded: in bounds
                }
            }
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            key[j] = ~0ull;
            if (j < R) {
                const u32 q = woff + j * WAVE;
                if (q < A) {
                    const u64 start = (u64)s_idx[q] + hsum[j];       // <= n: the group shares that many real characters
                    const u64 avail = a.n - start;
                    const u64 w = wtext[j];
                    u32 kcg = kc0;
                    if (a.L) { const u32 rem = a.L - hsum[j]; kcg = rem < kcg ? rem : kcg; }   // hsum < L while a record is active
                    const u32 sig = kcg * ub;
                    const u32 w0 = (u32)w, w1 = (u32)(w >> 32);
                    const u32 c0 = s_map[w0 & 255u], c1 = s_map[(w0 >> 8) & 255u], c2 = s_map[(w0 >> 16) & 255u], c3 = s_map[w0 >> 24];
                    u64 f;
                    if (ub <= 8) {   // uniform
                        const u32 c4 = s_map[w1 & 255u], c5 = s_map[(w1 >> 8) & 255u], c6 = s_map[(w1 >> 16) & 255u], c7 = s_map[w1 >> 24];
                        const u32 hi4 = (((((c0 << ub) | c1) << ub) | c2) << ub) | c3;
                        const u32 lo4 = (((((c4 << ub) | c5) << ub) | c6) << ub) | c7;
                        u64 all = ((u64)hi4 << (4 * ub)) | lo4;          // character 0 on top, 8 * b <= 64 bits
                        if (avail < 8) all = avail ? (all & (~0ull << (ub * (8 - (u32)avail)))) : 0ull;
                        f = (8 * ub >= (u32)FIN_FIELD) ? (all >> (8 * ub - (u32)FIN_FIELD)) : (all << ((u32)FIN_FIELD - 8 * ub));
                    } else {         // b == 9: four characters
                        u64 hi = ((((((u64)c0 << ub) | c1) << ub) | c2) << ub) | c3;
                        if (avail < 4) hi = avail ? (hi & (~0ull << (ub * (4 - (u32)avail)))) : 0ull;
                        f = hi << ((u32)FIN_FIELD - 4 * ub);
                    }
                    f &= ~0ull << ((u32)FIN_FIELD - sig);               // characters beyond kc (and L) do not count
                    key[j] = (f << (2 * FIN_POS_BITS)) | ((u64)q << FIN_POS_BITS);
                    s_key[q] = key[j];
                }
            }
        }
        if (a.debug) { const long long t = clock64(); t_fetch += t - t_mark; t_mark = t; }
        sync_lds();   // keys in place; s_big / s_nbig complete (LDS atomics above)
        const u32 nbig = (u32)__builtin_amdgcn_readfirstlane((int)s_nbig);
        if (nbig) { ++dbg_bigrounds; }

        // 2a. small groups: a record's place inside its group = the members with a smaller key (keys are distinct: they
        //     carry the compact index, which also keeps ties in text order)
        u32 dst[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            dst[j] = 0;
            if (j < R) {
                const u32 q = woff + j * WAVE;
                if (q < A) {
                    const u32 g = s_lgid[q];
                    const u32 q0 = s_gqp[g] & 0xFFFFu, q1 = s_gqp[g + 1] & 0xFFFFu;
                    if (q1 - q0 <= cm) {
                        u32 rank = 0;
                        const u64 kj = key[j];
                        for (u32 m = q0; m < q1; m += 4) {
                            const u64 k0 = s_key[m], k1 = s_key[(m + 1) & PM], k2 = s_key[(m + 2) & PM], k3 = s_key[(m + 3) & PM];
                            rank += (k0 < kj ? 1u : 0u) + ((m + 1 < q1 && k1 < kj) ? 1u : 0u) + ((m + 2 < q1 && k2 < kj) ? 1u : 0u) +
                                    ((m + 3 < q1 && k3 < kj) ? 1u : 0u);
                        }
                        dst[j] = q0 + rank;
                    } else dst[j] = ~0u;   // its place comes from the wave that splits the group
                }
            }
        }
        // 2b. large groups, one wave each (no workgroup barrier inside); four rows of 64 members in flight per step
        for (u32 bi = (u32)wave; bi < nbig; bi += WAVES) {
            const u32 g = s_big[bi];
            const u32 q0 = s_gqp[g] & 0xFFFFu, q1 = s_gqp[g + 1] & 0xFFFFu, m = q1 - q0;
            dbg_bigrec += m;
            u32 kcg = kc0;
            if (a.L) { const u32 rem = a.L - (a.h0 + (u32)s_gh[g]); kcg = rem < kcg ? rem : kcg; }
            const u32 sigmax = kcg * ub;
            const u64 f0 = s_key[q0] >> (2 * FIN_POS_BITS);
            u64 x = 0;
            for (u32 r0 = 0; r0 < m; r0 += 4 * WAVE) {
                u64 kk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const u32 r = r0 + u * WAVE + lane; kk[u] = s_key[q0 + (r < m ? r : 0u)]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) x |= (kk[u] >> (2 * FIN_POS_BITS)) ^ f0;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x |= __shfl_xor(x, o);
            const u32 lcp = x ? (u32)__clzll((long long)x) - (u32)(64 - FIN_FIELD) : (u32)FIN_FIELD;
            u32 sig;
            if (lcp >= sigmax) {
                // every member has the same kc characters: nothing to order, the group moves on as it is
                sig = sigmax;
                for (u32 r = lane; r < m; r += 64) {
                    u32* lo = reinterpret_cast<u32*>(&s_key[q0 + r]);
                    *lo = (*lo & ~PM) | (q0 + r);
                }
            } else {
                const u32 W = (sigmax - lcp) < (u32)FIN_SPLIT ? (sigmax - lcp) : (u32)FIN_SPLIT;
                const u32 shift = (u32)(2 * FIN_POS_BITS) + (u32)FIN_FIELD - lcp - W;
                const u32 mask = (1u << W) - 1u;
                u32* wh32 = reinterpret_cast<u32*>(wh);   // two 16-bit digit counters per word (a group has at most CAP members)
                for (u32 i2 = lane; i2 < (1u << FIN_SPLIT) / 2; i2 += 64) wh32[i2] = 0;
                __builtin_amdgcn_wave_barrier();
                for (u32 r0 = 0; r0 < m; r0 += 4 * WAVE) {
                    u64 kk[4]; u32 dd[4], below[4], total[4], lead[4], prior[4];
                    bool valid[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u32 r = r0 + u * WAVE + lane;
                        valid[u] = r < m;
                        kk[u] = s_key[q0 + (valid[u] ? r : 0u)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        dd[u] = valid[u] ? ((u32)(kk[u] >> shift) & mask) : 0u;
                        u32 plo = ~0u, phi = ~0u;
#pragma unroll
                        for (int bb = 0; bb < FIN_SPLIT; ++bb) {
                            u32 e = (u32)__builtin_amdgcn_sbfe((int)dd[u], bb, 1);
                            asm("" : "+v"(e));
                            const u64 mm = __ballot(e != 0);
                            plo &= ~((u32)mm ^ e);
                            phi &= ~((u32)(mm >> 32) ^ e);
                        }
                        const u64 vm = __ballot(valid[u]);
                        plo &= (u32)vm; phi &= (u32)(vm >> 32);
                        below[u] = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
                        total[u] = (u32)__popc(plo) + (u32)__popc(phi);
                        lead[u] = plo ? (u32)__builtin_ctz(plo) : 32u + (u32)__builtin_ctz(phi | 0x80000000u);   // a valid lane is its own peer
                    }
                    // the first lane of every digit adds its row's count and learns how many members of earlier rows carry the digit
                    // (the returning atomics of one wave are served in program order: row after row, stable)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        prior[u] = 0;
                        if (valid[u] && below[u] == 0)
                            prior[u] = (atomicAdd(&wh32[dd[u] >> 1], total[u] << (16u * (dd[u] & 1u))) >> (16u * (dd[u] & 1u))) & 0xFFFFu;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u32 pr = (u32)__shfl((int)prior[u], (int)lead[u]);
                        if (valid[u]) {
                            u32* lo32 = reinterpret_cast<u32*>(&s_key[q0 + r0 + u * WAVE + lane]);
                            *lo32 = ((u32)kk[u] & ~PM) | (pr + below[u]);   // rank among the members with this digit
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                {   // exclusive prefix of the digit counts
                    const u32 bins = 1u << W;
                    const u32 per = bins >= 64u ? bins / 64u : 1u;
                    constexpr u32 PER_MAX = (1 << FIN_SPLIT) / 64 > 0 ? (1 << FIN_SPLIT) / 64 : 1;
                    u32 v[PER_MAX];
                    u32 sum = 0;
#pragma unroll
                    for (u32 e = 0; e < PER_MAX; ++e) {
                        const u32 bin = (u32)lane * per + e;
                        v[e] = (e < per && bin < bins) ? (u32)wh[bin] : 0u;
                        sum += v[e];
                    }
                    u32 incl = sum;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const u32 t = __shfl_up(incl, o);
                        if (lane >= o) incl += t;
                    }
                    u32 run = incl - sum;
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (u32 e = 0; e < PER_MAX; ++e) {
                        const u32 bin = (u32)lane * per + e;
                        if (e < per && bin < bins) { wh[bin] = (u16)run; run += v[e]; }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                for (u32 r0 = 0; r0 < m; r0 += 4 * WAVE) {
                    u64 kk[4]; u32 base[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const u32 r = r0 + u * WAVE + lane; kk[u] = s_key[q0 + (r < m ? r : 0u)]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) base[u] = wh[(u32)(kk[u] >> shift) & mask];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u32 r = r0 + u * WAVE + lane;
                        if (r < m) {
                            u32* lo32 = reinterpret_cast<u32*>(&s_key[q0 + r]);
                            *lo32 = ((u32)kk[u] & ~PM) | (q0 + base[u] + ((u32)kk[u] & PM));
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                sig = lcp + W;
            }
            if (lane == 0) s_gsig[g] = (u8)sig;
        }
        if (a.debug) { const long long t = clock64(); t_sort += t - t_mark; t_mark = t; }
        __syncthreads();   // every place is known; the keys of the small groups have been read
        if (nbig) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                if (j < R) {
                    const u32 q = woff + j * WAVE;
                    if (q < A && dst[j] == ~0u) { const u64 k = s_key[q]; dst[j] = (u32)k & PM; key[j] = k; }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (j < R) { const u32 q = woff + j * WAVE; if (q < A) s_key[dst[j]] = key[j]; }
        }
        __syncthreads();
        if (a.debug) { const long long t = clock64(); t_perm += t - t_mark; t_mark = t; }

        // 3. new groups, finals (straight to their SA slots), compaction of what is still tied
        u32 r_idx[ITEMS], r_ph[ITEMS];   // r_ph: position in the tile | new depth << 16
        u64 m_act[ITEMS], m_head[ITEMS];
        u32 finmask = 0, headmask = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            m_act[j] = 0; m_head[j] = 0; r_idx[j] = 0; r_ph[j] = 0;
            if (j < R) {
                const u32 q = woff + j * WAVE;
                const bool valid = q < A;
                bool head = false, fin = false;
                if (valid) {
                    const u32 g = s_lgid[q];                        // q's group is the one it had before: sorting stays inside groups
                    const u32 gqp = s_gqp[g];
                    const u32 q0 = gqp & 0xFFFFu, q1 = s_gqp[g + 1] & 0xFFFFu;
                    const u32 gh = s_gh[g];
                    u32 sig = s_gsig[g];
                    if (sig == 0) {
                        u32 kcg = kc0;
                        if (a.L) { const u32 rem = a.L - (a.h0 + gh); kcg = rem < kcg ? rem : kcg; }
                        sig = kcg * ub;
                    }
                    const u32 cs = (u32)(2 * FIN_POS_BITS) + (u32)FIN_FIELD - sig;   // sig >= 1
                    const u64 kq = s_key[q];
                    const u64 k = kq >> cs;
                    const u64 kp = s_key[(q - 1) & PM] >> cs, kn = s_key[(q + 1) & PM] >> cs;
                    head = (q == q0) || (k != kp);
                    const bool tail = (q + 1 == q1) || (kn != k);
                    const u32 newh = gh + ((sig * inv_b) >> 16);
                    fin = (head && tail) || (a.L && a.h0 + newh >= a.L);
                    r_idx[j] = s_idx[(u32)(kq >> FIN_POS_BITS) & PM];   // the record that now stands at q
                    r_ph[j] = ((gqp >> 16) + (q - q0)) | (newh << 16);
                }
                if (fin) { finmask |= 1u << j; if (head) headmask |= 1u << j; }
                const bool act = valid && !fin;
                m_act[j] = __ballot(act);
                m_head[j] = __ballot(act && head);
                if (lane == 0) { s_rowa[wave * ITEMS + j] = (u32)__popcll(m_act[j]); s_rowh[wave * ITEMS + j] = (u32)__popcll(m_head[j]); }
            }
        }
        // finals: the slot loads of a lane are requested here, the stores follow the compaction below
        u32 slotv[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            slotv[j] = 0;
            if (j < R && ((finmask >> j) & 1u)) slotv[j] = a.apos[begin + (r_ph[j] & 0xFFFFu)];
        }
        __syncthreads();   // every read of s_key / s_idx / s_lgid / s_gqp / s_gh / s_gsig of this round is done
        u32 ia = s_rowa[lane], ih = s_rowh[lane];   // every wave scans the 64 row counts
        const u32 ca = ia, ch = ih;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 ta = __shfl_up(ia, o), th = __shfl_up(ih, o);
            if (lane >= o) { ia += ta; ih += th; }
        }
        const u32 tot_a = (u32)__builtin_amdgcn_readlane((int)ia, 63), tot_h = (u32)__builtin_amdgcn_readlane((int)ih, 63);
        const u32 ea = ia - ca, eh = ih - ch;
        const u64 lt_mask = lanemask_lt();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (j < R) {
                const u32 row = (u32)wave * ITEMS + j;
                const u32 base_a = __shfl(ea, (int)row), base_h = __shfl(eh, (int)row);
                if ((m_act[j] >> lane) & 1ull) {
                    const u32 na = base_a + (u32)__popcll(m_act[j] & lt_mask);
                    const u32 ng = base_h + (u32)__popcll(m_head[j] & (lt_mask | (1ull << lane))) - 1u;
                    s_idx[na] = r_idx[j];
                    s_lgid[na] = (u16)ng;
                    if ((m_head[j] >> lane) & 1ull) {
                        s_gqp[ng] = na | ((r_ph[j] & 0xFFFFu) << 16);
                        s_gh[ng] = (u8)(r_ph[j] >> 16);
                        s_gsig[ng] = 0;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (j < R && ((finmask >> j) & 1u)) {
                a.sa[slotv[j]] = r_idx[j];
                if ((headmask >> j) & 1u) a.gflags[slotv[j]] = 1;
                a.done[begin + (r_ph[j] & 0xFFFFu)] = 1;
                ++resolved;
            }
        }
        stall = (tot_a < A || tot_h > G) ? 0u : stall + 1u;
        if (tid == 0) { s_A = tot_a; s_G = tot_h; s_gqp[tot_h] = tot_a; s_nbig = 0; }
        __syncthreads();
        ++rounds;
        if (a.debug) { const long long t = clock64(); t_regroup += t - t_mark; t_mark = t; }
    }
    sync_lds();   // s_fail / s_anyfail complete; the finals of every lane have reached memory (workgroup scope)
    if (__builtin_amdgcn_readfirstlane((int)s_anyfail)) {
        // a group the tile gave up on goes back to the global path EXACTLY as it was: the slots of its members that had
        // already been written get their suffix of the list back, the head marks inside it are taken back
        for (u32 p = tid; p < cnt; p += FIN_BLOCK) {
            const u32 gp = a.gid[begin + p];
            const u32 og = gp - gid0;
            if (((s_fail[og >> 5] >> (og & 31)) & 1u) && a.done[begin + p]) {
                const u32 slot = a.apos[begin + p];
                a.sa[slot] = a.aidx[begin + p];
                a.gflags[slot] = (p == 0 || a.gid[begin + p - 1] != gp) ? (u8)1 : (u8)2;   // bit0 = head of a group
                a.done[begin + p] = 0;
                --resolved;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) resolved += __shfl_down(resolved, o);
    if (lane == 0 && resolved) atomicAdd(&a.totals[1], (unsigned long long)(long long)(int)resolved);
    if (tid == 0) {
        atomicAdd(&a.totals[0], (unsigned long long)cnt);
        if (a.debug) {
            atomicAdd(&a.totals[2], (unsigned long long)rounds); atomicAdd(&a.totals[3], (unsigned long long)dbg_bigrounds);
            atomicAdd(&a.totals[4], (unsigned long long)dbg_bigrec); atomicAdd(&a.totals[5], (unsigned long long)dbg_slots);
            atomicAdd(&a.totals[6], 1ull); atomicAdd(&a.totals[7], (unsigned long long)dbg_act);
            atomicAdd(&a.totals[8], (unsigned long long)t_fetch); atomicAdd(&a.totals[9], (unsigned long long)t_sort);
            atomicAdd(&a.totals[10], (unsigned long long)t_perm); atomicAdd(&a.totals[11], (unsigned long long)t_regroup);
            atomicAdd(&a.totals[12], (unsigned long long)(clock64() - t_mark));
        }
    }
}


}  // namespace sa
