/*
 * sa_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (suffixarray_amd/) never links, imports or calls it.
 *
 * What is restated (reference file:line, relative to /root/reference):
 *   C1/C2  suffix_array/libsais.c:6618-6632 (libsais), 6480-6519 (libsais_main_8u):
 *          SA-IS (Nong/Zhang/Chan induced sorting) -> oracle_sais().  libsais is the
 *          vendored third-party library (libsais 2.8.4); this file restates the
 *          *published algorithm* in ~150 lines, not libsais' blocked/OpenMP code.
 *   C3     suffix_array/libsais64.c:6657-6685: for n <= INT32_MAX run the 32-bit sort
 *          in the low half of the int64 buffer, then widen back-to-front -> oracle_sais64().
 *   C4     suffix_array/engine.c:696-812,837-866 (construct_truncated_suffix_array):
 *          MSD 256-way counting sort per depth, depth capped at L -> oracle_truncated_sa().
 *          The reference's insertion-sort shortcut (engine.c:710-729) and its skipped
 *          '\n'/0xFF buckets (engine.c:771-772) make its output non-canonical; the
 *          restatement is the exact "stable sort by the first L bytes" contract.
 *   Q1     suffix_array/engine.c:869-918 (get_substring_positions) -> oracle_get_substring_positions().
 *
 * Pinning: the reference holds no golden vectors for this path, so this
 * restatement is pinned against the reference itself compiled in place (oracle/_ref,
 * see oracle/Makefile) by tests/test_oracle.py, and against fixtures generated
 * from that build (tests/golden/, generator tests/golden/make_golden.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * C1/C2: SA-IS.  Text symbols are read through CH() so the same code serves the byte text
 * (cs == 1) and the int32 reduced text of the recursion (cs == 4), as libsais_main_8u /
 * libsais_main_32s do (libsais.c:6480, 6269).  The implicit sentinel is smaller than every
 * symbol (libsais.c:703-707: the scan is seeded with c1 = -1), i.e. a suffix that is a proper
 * prefix of another sorts first, and bytes compare unsigned.
 * ---------------------------------------------------------------------------------------- */
#define CH(i) (cs == 1 ? (int32_t)((const uint8_t*)T)[(i)] : ((const int32_t*)T)[(i)])
#define IS_S(i) (types[(i)])
#define IS_LMS(i) ((i) > 0 && types[(i)] && !types[(i) - 1])

static void bucket_bounds(const int32_t* C, int32_t* B, int32_t K, int end) {
    int32_t sum = 0;
    for (int32_t c = 0; c < K; ++c) { sum += C[c]; B[c] = end ? sum : sum - C[c]; }
}

/* Induce L-type suffixes left-to-right, then S-type right-to-left
 * (libsais.c:3813 induce_partial_order / 5814 induce_final_order, scans 2110 and 2942). */
static void induce(const void* T, int32_t* SA, int32_t n, int32_t K, int cs,
                   const uint8_t* types, const int32_t* C, int32_t* B) {
    bucket_bounds(C, B, K, 0);
    /* the (virtual) sentinel suffix n is first in SA and induces suffix n-1, which is L-type */
    SA[B[CH(n - 1)]++] = n - 1;
    for (int32_t i = 0; i < n; ++i) {
        int32_t j = SA[i];
        if (j > 0 && !IS_S(j - 1)) SA[B[CH(j - 1)]++] = j - 1;
    }
    bucket_bounds(C, B, K, 1);
    for (int32_t i = n - 1; i >= 0; --i) {
        int32_t j = SA[i];
        if (j > 0 && IS_S(j - 1)) SA[--B[CH(j - 1)]] = j - 1;
    }
}

static int sais_rec(const void* T, int32_t* SA, int32_t n, int32_t K, int cs) {
    if (n == 0) return 0;
    if (n == 1) { SA[0] = 0; return 0; }
    uint8_t* types = (uint8_t*)malloc((size_t)n);
    int32_t* C = (int32_t*)calloc((size_t)K, sizeof(int32_t));
    int32_t* B = (int32_t*)malloc((size_t)K * sizeof(int32_t));
    if (!types || !C || !B) { free(types); free(C); free(B); return -2; }

    /* S/L classification, right to left (libsais.c:693-804 count_and_gather_lms_suffixes) */
    types[n - 1] = 0;
    for (int32_t i = n - 2; i >= 0; --i) {
        int32_t a = CH(i), b = CH(i + 1);
        types[i] = (uint8_t)((a < b) || (a == b && types[i + 1]));
    }
    for (int32_t i = 0; i < n; ++i) C[CH(i)]++;

    /* stage 1: sort the LMS substrings (libsais.c:1542 radix_sort_lms + 3813 induce_partial_order) */
    for (int32_t i = 0; i < n; ++i) SA[i] = -1;
    bucket_bounds(C, B, K, 1);
    for (int32_t i = 1; i < n; ++i) if (IS_LMS(i)) SA[--B[CH(i)]] = i;
    induce(T, SA, n, K, cs, types, C, B);

    /* gather sorted LMS substrings and name them (libsais.c:3860-4042 renumber_and_gather) */
    int32_t m = 0;
    for (int32_t i = 0; i < n; ++i) { int32_t j = SA[i]; if (IS_LMS(j)) SA[m++] = j; }
    for (int32_t i = m; i < n; ++i) SA[i] = -1;
    int32_t names = 0, prev = -1;
    for (int32_t i = 0; i < m; ++i) {
        int32_t pos = SA[i];
        int diff = (prev < 0);
        for (int32_t d = 0; !diff; ++d) {
            if (pos + d >= n || prev + d >= n) { diff = 1; break; }
            if (CH(pos + d) != CH(prev + d) || types[pos + d] != types[prev + d]) { diff = 1; break; }
            if (d > 0 && (IS_LMS(pos + d) || IS_LMS(prev + d))) {
                if (!(IS_LMS(pos + d) && IS_LMS(prev + d))) diff = 1;
                break;
            }
        }
        if (diff) { ++names; prev = pos; }
        SA[m + (pos >> 1)] = names - 1;
    }
    int32_t j = n - 1;
    for (int32_t i = n - 1; i >= m; --i) if (SA[i] >= 0) SA[j--] = SA[i];
    int32_t* s1 = SA + n - m;

    /* stage 2: order of the LMS suffixes = SA of the reduced string (libsais.c:6269 recursion) */
    int rc = 0;
    if (names < m) {
        rc = sais_rec(s1, SA, m, names, 4);
    } else {
        for (int32_t i = 0; i < m; ++i) SA[s1[i]] = i;
    }
    if (rc != 0) { free(types); free(C); free(B); return rc; }

    /* stage 3: place sorted LMS suffixes at their bucket ends, final induce
     * (libsais.c:4305-4377 reconstruct/place, 5814 induce_final_order) */
    j = 0;
    for (int32_t i = 1; i < n; ++i) if (IS_LMS(i)) s1[j++] = i;
    for (int32_t i = 0; i < m; ++i) SA[i] = s1[SA[i]];
    for (int32_t i = m; i < n; ++i) SA[i] = -1;
    bucket_bounds(C, B, K, 1);
    for (int32_t i = m - 1; i >= 0; --i) {
        int32_t p = SA[i];
        SA[i] = -1;
        SA[--B[CH(p)]] = p;
    }
    induce(T, SA, n, K, cs, types, C, B);

    free(types); free(C); free(B);
    return 0;
}

/* libsais-compatible: 0 ok, -1 bad args, -2 out of memory (libsais.h:82-94); freq = byte
 * histogram when non-NULL (libsais.c:1363-1371). */
int32_t oracle_sais(const uint8_t* T, int32_t* SA, int32_t n, int32_t fs, int32_t* freq) {
    if (T == NULL || SA == NULL || n < 0 || fs < 0) return -1;
    if (freq != NULL) {
        memset(freq, 0, 256 * sizeof(int32_t));
        for (int32_t i = 0; i < n; ++i) freq[T[i]]++;
    }
    if (n < 2) { if (n == 1) SA[0] = 0; return 0; }
    return sais_rec(T, SA, n, 256, 1);
}

/* C3: libsais64.c:6657-6685 -- n <= INT32_MAX: 32-bit sort in the low half, widen in place
 * back to front (libsais64.c:6248-6259). */
int64_t oracle_sais64(const uint8_t* T, int64_t* SA, int64_t n, int64_t fs, int64_t* freq) {
    if (T == NULL || SA == NULL || n < 0 || fs < 0) return -1;
    if (n > INT32_MAX) return -1; /* beyond the 32-bit path: not restated */
    if (freq != NULL) {
        memset(freq, 0, 256 * sizeof(int64_t));
        for (int64_t i = 0; i < n; ++i) freq[T[i]]++;
    }
    if (n < 2) { if (n == 1) SA[0] = 0; return 0; }
    int32_t* lo = (int32_t*)SA;
    int rc = sais_rec(T, lo, (int32_t)n, 256, 1);
    if (rc != 0) return rc;
    for (int64_t i = n - 1; i >= 0; --i) SA[i] = (int64_t)(uint32_t)lo[i];
    return 0;
}

/* Definition of the suffix array, for tiny inputs only: qsort with the suffix comparator
 * (unsigned bytes, shorter suffix first). */
static const uint8_t* g_T; static uint32_t g_n;
static int cmp_suffix(const void* a, const void* b) {
    uint32_t i = *(const uint32_t*)a, j = *(const uint32_t*)b;
    uint32_t li = g_n - i, lj = g_n - j, l = li < lj ? li : lj;
    int c = memcmp(g_T + i, g_T + j, l);
    if (c != 0) return c;
    return (li < lj) ? -1 : (li > lj);
}
void oracle_sa_naive(const uint8_t* T, uint32_t n, uint32_t* SA) {
    for (uint32_t i = 0; i < n; ++i) SA[i] = i;
    g_T = T; g_n = n;
    qsort(SA, n, sizeof(uint32_t), cmp_suffix);
}

/* ------------------------------------------------------------------------------------------
 * C4: truncated suffix array.  Same shape as engine.c:696-812 (identity init engine.c:845-848,
 * then per depth: 256-bin histogram, exclusive scan, stable scatter through a temp array,
 * recurse into every bucket) but exact: depth stops at L = min(max_suffix_length, n)
 * (engine.c:841), a suffix that ends inside the bucket sorts first (strncmp against the
 * terminating NUL, engine.c:886), every bucket is refined, no insertion-sort shortcut.
 * Result: suffixes ordered by their first L bytes; ties keep text order (stable).
 * ---------------------------------------------------------------------------------------- */
static void msd_sort(const uint8_t* T, uint32_t n, uint32_t* sa, uint32_t* tmp,
                     uint32_t cnt, uint32_t depth, uint32_t L) {
    while (cnt > 1 && depth < L) {
        uint32_t hist[257];
        memset(hist, 0, sizeof hist);
        /* bin 0 = suffix ended (sa[i] + depth == n), bin c+1 = byte c */
        for (uint32_t i = 0; i < cnt; ++i) {
            uint64_t p = (uint64_t)sa[i] + depth;
            hist[p >= n ? 0 : (uint32_t)T[p] + 1]++;
        }
        uint32_t start[258], off = 0, nonempty = 0, only = 0;
        for (uint32_t b = 0; b < 257; ++b) { start[b] = off; off += hist[b]; if (hist[b]) { ++nonempty; only = b; } }
        start[257] = off;
        if (nonempty == 1) { if (only == 0) return; ++depth; continue; }
        uint32_t cur[257];
        memcpy(cur, start, sizeof cur);
        for (uint32_t i = 0; i < cnt; ++i) {
            uint64_t p = (uint64_t)sa[i] + depth;
            tmp[cur[p >= n ? 0 : (uint32_t)T[p] + 1]++] = sa[i];
        }
        memcpy(sa, tmp, (size_t)cnt * sizeof(uint32_t));
        for (uint32_t b = 1; b < 257; ++b)
            if (hist[b] > 1) msd_sort(T, n, sa + start[b], tmp + start[b], hist[b], depth + 1, L);
        return;
    }
}
void oracle_truncated_sa(const uint8_t* T, uint32_t n, uint32_t max_suffix_length, uint32_t* SA) {
    uint32_t L = max_suffix_length < n ? max_suffix_length : n;
    for (uint32_t i = 0; i < n; ++i) SA[i] = i;
    if (n < 2) return;
    uint32_t* tmp = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    msd_sort(T, n, SA, tmp, n, 0, L);
    free(tmp);
}

/* ------------------------------------------------------------------------------------------
 * Q1: engine.c:869-918.  Two binary searches over SA with strncmp(str + SA[mid], q, c),
 * c = min(strlen(q), max_suffix_length) (engine.c:881).  The text is NUL-terminated and q has
 * no NUL in its first c bytes, so strncmp == "memcmp over the suffix, a suffix shorter than c
 * that matches so far compares less" -- the form used here, which is also defined for texts
 * containing 0x00 bytes.
 * Result conventions (engine.c:896-898,916-917): {first,last} inclusive; no suffix >= q ->
 * {UINT32_MAX,UINT32_MAX}; miss -> first = lower bound, last = first-1.
 * NOT reproduced: the reference computes `last = mid - 1` in uint32 (engine.c:891,908); with
 * mid == 0 that wraps and the loop reads out of bounds.  Indices are signed 64-bit here, so
 * q <= smallest suffix yields the natural {0, ub-1} (ub = 0 -> last = UINT32_MAX).
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t first, second; } oracle_pair_u32;

static inline int cmp_suffix_pat(const uint8_t* T, uint64_t n, uint64_t pos,
                                 const uint8_t* q, uint32_t c) {
    uint64_t avail = n - pos;
    uint32_t l = avail < c ? (uint32_t)avail : c;
    int r = memcmp(T + pos, q, l);
    if (r != 0) return r;
    return l < c ? -1 : 0;
}

oracle_pair_u32 oracle_get_substring_positions(const uint8_t* T, uint64_t n, const uint32_t* SA,
                                               uint32_t max_suffix_length,
                                               const uint8_t* q, uint32_t m) {
    uint32_t c = m < max_suffix_length ? m : max_suffix_length;
    int64_t first = 0, last = (int64_t)n - 1;
    int64_t start = -1, end = -1;
    while (first <= last) {
        int64_t mid = (first + last) / 2;
        if (cmp_suffix_pat(T, n, SA[mid], q, c) < 0) first = mid + 1;
        else { last = mid - 1; start = mid; }
    }
    oracle_pair_u32 r = { UINT32_MAX, UINT32_MAX };
    if (start < 0) return r;
    first = 0; last = (int64_t)n - 1;
    while (first <= last) {
        int64_t mid = (first + last) / 2;
        if (cmp_suffix_pat(T, n, SA[mid], q, c) > 0) last = mid - 1;
        else { first = mid + 1; end = mid; }
    }
    r.first = (uint32_t)start;
    r.second = end < 0 ? UINT32_MAX : (uint32_t)end;
    return r;
}

/* Batched form (new surface of the build; reduces to Q1 per element).  Patterns are packed
 * back to back, pattern i = pat[off[i] .. off[i+1]).  OpenMP over the batch as SURVEY 8(d)
 * prescribes for the CPU query baseline.  Returns the thread count used. */
int oracle_query_batch(const uint8_t* T, uint64_t n, const uint32_t* SA, uint32_t max_suffix_length,
                       const uint8_t* pat, const uint64_t* off, uint64_t Q,
                       oracle_pair_u32* out, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
    #pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
#endif
    for (int64_t i = 0; i < (int64_t)Q; ++i)
        out[i] = oracle_get_substring_positions(T, n, SA, max_suffix_length,
                                                pat + off[i], (uint32_t)(off[i + 1] - off[i]));
    (void)threads;
    return used;
}

/* O(n) suffix-array checker (SURVEY 8c "sufcheck"): permutation + first-byte order + rank of
 * the successor suffix as tie-break.  Returns 0 when SA is the suffix array of T. */
int oracle_sufcheck(const uint8_t* T, uint64_t n, const uint32_t* SA) {
    if (n == 0) return 0;
    uint32_t* isa = (uint32_t*)malloc(n * sizeof(uint32_t));
    if (!isa) return -2;
    memset(isa, 0xFF, n * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) {
        if (SA[i] >= n || isa[SA[i]] != UINT32_MAX) { free(isa); return 1; }
        isa[SA[i]] = (uint32_t)i;
    }
    for (uint64_t i = 1; i < n; ++i) {
        uint32_t a = SA[i - 1], b = SA[i];
        if (T[a] > T[b]) { free(isa); return 2; }
        if (T[a] == T[b]) {
            int64_t ra = (a + 1 < n) ? (int64_t)isa[a + 1] : -1;
            int64_t rb = (b + 1 < n) ? (int64_t)isa[b + 1] : -1;
            if (ra >= rb) { free(isa); return 3; }
        }
    }
    free(isa);
    return 0;
}
