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

}  // namespace sa
