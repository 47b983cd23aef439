/*
 * sa_hip.h -- C ABI of libsa_hip.so: MI355X (gfx950) suffix-array construction and batched
 * substring query.  Plain C types only; no C++, HIP or torch types cross this boundary
 * (device pointers and the stream travel as void*).
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference repository jdm365/SuffixArray @ 2024_10_08).  INTEGRATION.md shows the
 * reference-side binding (Cython `cdef extern`, Makefile link line).
 *
 * Error convention: libsais' (libsais.h:82-94) -- 0 ok, -1 invalid arguments, -2 out of
 * (host or device) memory; additionally -3 HIP runtime / no usable device, -4 internal
 * device-side failure (bounded spin expired).  Nothing here calls exit() or prints.
 * sa_hip_last_error() returns a thread-local message for the last non-zero return.
 *
 * Threading: every function may be called without the GIL.  A handle owns one HIP stream;
 * calls on the same handle are serialised by an internal mutex; different handles are
 * independent.  The libsais-/engine-compatible wrappers create a private handle per call.
 */
#ifndef SA_HIP_H
#define SA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_HIP_OK            0
#define SA_HIP_EINVAL       (-1)
#define SA_HIP_ENOMEM       (-2)
#define SA_HIP_EHIP         (-3)
#define SA_HIP_EINTERNAL    (-4)

/* ---- ABI structs (layout identical to the reference's) --------------------------------- */

/* engine.h:219-222 */
typedef struct sa_hip_pair_u32 {
    uint32_t first;
    uint32_t second;
} sa_hip_pair_u32;

/* engine.h:123-130 SuffixArray_struct; sizeof == 40.  is_quoted_bitflag is opaque here. */
typedef struct sa_hip_SuffixArray_struct {
    uint32_t* suffix_array;
    void*     is_quoted_bitflag;
    uint64_t  global_byte_start_idx;
    uint64_t  global_byte_end_idx;
    uint32_t  max_suffix_length;
    uint32_t  n;
} sa_hip_SuffixArray_struct;

/* ---- (1) construction, libsais-call-compatible (host pointers in, host SA out) ---------- */

/* replaces libsais (libsais.h:84, libsais.c:6618).  SA[0..n) <- suffix array of T[0..n);
 * SA[n..n+fs) untouched; freq (if non-NULL) <- 256-bin byte histogram. */
int32_t sa_hip_libsais(const uint8_t* T, int32_t* SA, int32_t n, int32_t fs, int32_t* freq);
/* replaces libsais_omp (libsais.h:121, libsais.c:6791).  threads is validated (>= 0) and
 * otherwise ignored: the device pipeline has no host thread pool. */
int32_t sa_hip_libsais_omp(const uint8_t* T, int32_t* SA, int32_t n, int32_t fs, int32_t* freq, int32_t threads);
/* replaces libsais64 (libsais64.h:61, libsais64.c:6657) for n <= UINT32_MAX - 1: 32-bit device
 * build + widening (the reference does the same on the CPU for n <= INT32_MAX, libsais64.c:6670-6682).
 * n > 2^32 - 2 (round 4): the counterpart of the reference's true 64-bit path (libsais64.c:6684 -> libsais64_main) --
 * 64-bit suffix indices on the device (csrc/big_build.hpp: radix sort of (u64 key, u64 suffix) records + prefix doubling
 * on what stays tied), 41 bytes of HBM per character: texts up to about 6.5e9 bytes on one 288 GB GPU, SA_HIP_ENOMEM (-2)
 * beyond.  A functional completion, outside every BASELINE configuration: plain (unpinned) copies, no shared workspace. */
int64_t sa_hip_libsais64(const uint8_t* T, int64_t* SA, int64_t n, int64_t fs, int64_t* freq);
/* replaces libsais64_omp (libsais64.h:86, libsais64.c:6783). */
int64_t sa_hip_libsais64_omp(const uint8_t* T, int64_t* SA, int64_t n, int64_t fs, int64_t* freq, int64_t threads);

/* The 64-bit-index build on device buffers (what sa_hip_libsais64 runs for n > 2^32 - 2; any n >= 0 is accepted, which is how
 * the tests compare it with the oracle at small sizes): text_dev = n bytes, 16-byte aligned; sa_dev = n int64 entries
 * (libsais64 layout).  No index handle is involved: the array is the product. */
typedef struct sa_hip_big_stats {
    uint32_t sigma, bits_per_symbol, initial_chars;   /* alphabet, bits per character code, characters in the initial key  */
    uint32_t sort_passes;                             /* 8-bit radix passes over (u64, u64) records, all sorts of the build */
    uint32_t rounds;                                  /* prefix-doubling rounds                                             */
    uint32_t pad_;
    uint64_t tied_after_sort;                         /* suffixes still tied after the initial sort                         */
    uint64_t tied_total;                              /* sum over the rounds of the tied suffixes they looked at            */
    float    total_ms;                                /* device time of the build (HIP events)                              */
} sa_hip_big_stats;
int sa_hip_libsais64_device(const void* text_dev, int64_t* sa_dev, int64_t n, int device, sa_hip_big_stats* stats /* or NULL */);
/* sufcheck with 64-bit indices: *violations = slots at which sa_dev is not a permutation of [0, n) in suffix order
 * (0 <=> it is THE suffix array of text_dev; 8 n bytes of scratch) */
int sa_hip_sufcheck64_device(const void* text_dev, const int64_t* sa_dev, int64_t n, int device, uint64_t* violations);

/* The four calls above (and sa_hip_construct_truncated_suffix_array) share ONE process-level workspace: a device index
 * whose buffers are allocated on the first call and grow on demand (about 18 bytes of HBM per character of the
 * largest text seen) and 512 MB of pinned host slabs through which the text goes up and the suffix array comes down
 * (as u32 over PCIe, widened to int64 by host threads for the libsais64 forms).  Calls are serialised on it. */
typedef struct sa_hip_call_breakdown {
    uint64_t n;
    uint32_t workspace_reused;   /* 1: no device or pinned allocation in this call                           */
    uint32_t pad_;
    double   total_ms;           /* wall time of the call                                                   */
    double   workspace_ms;       /* creating / growing the workspace                                        */
    double   upload_ms;          /* text: pageable host memory -> pinned slabs -> HBM                        */
    double   build_ms;           /* wall time of the device build incl. its host synchronisations           */
    double   build_device_ms;    /* ... HIP-event time of the same                                          */
    double   download_ms;        /* suffix array: HBM -> pinned slabs -> (widened into) the caller's array  */
} sa_hip_call_breakdown;
/* Where the time of the last of those calls in this process went. */
int sa_hip_last_call_breakdown(sa_hip_call_breakdown* out);
/* Free the shared workspace (device buffers and pinned slabs); the next call allocates it again. */
void sa_hip_release_workspace(void);

/* ---- (2) truncated construction, engine.c-call-compatible ------------------------------- */

/* replaces construct_truncated_suffix_array (engine.h:213, engine.c:837-866).
 * Fills the caller-allocated sa->suffix_array[0..sa->n) with the suffixes of text[0..sa->n)
 * ordered by their first min(sa->max_suffix_length, n) bytes (unsigned, a suffix that ends
 * sorts first), ties in text order.  Returns 0 or a negative SA_HIP_* code (the reference
 * returns void and exit()s on failure). */
int sa_hip_construct_truncated_suffix_array(const char* text, sa_hip_SuffixArray_struct* sa);

/* ---- (3) query -------------------------------------------------------------------------- */

/* replaces get_substring_positions (engine.h:229-233, engine.c:869-918): one query, host
 * text (n bytes; a trailing NUL is not required) and host SA.  COMPATIBILITY SHIM, O(n) PER CALL: it creates a
 * private index, uploads text and SA and rebuilds the key array and the directory for ONE query.  Anything that
 * asks more than once must use the handle API below (sa_hip_index_load once, then sa_hip_query_batch). */
sa_hip_pair_u32 sa_hip_get_substring_positions(const char* str, const sa_hip_SuffixArray_struct* sa,
                                               const char* substring);

/* ---- (4) handle API: text + SA resident in HBM ------------------------------------------ */

typedef struct sa_hip_index sa_hip_index;

/* Number of HIP devices visible to this process, or a negative SA_HIP_* code. */
int sa_hip_device_count(void);

/* Create an empty index bound to `device` with capacity for texts of up to n_max bytes
 * (device workspace is allocated here, not in the build call). n_max <= UINT32_MAX - 1. */
int sa_hip_index_create(sa_hip_index** out, uint64_t n_max, int device);
void sa_hip_index_destroy(sa_hip_index* idx);

/* Build from a host text (H2D copy + device build).  max_suffix_length == 0: full suffix
 * array (libsais order); > 0: truncated order as in (2). */
int sa_hip_index_build(sa_hip_index* idx, const uint8_t* T_host, uint64_t n, uint32_t max_suffix_length);
/* Same, text already in device memory of idx's device (copied device-to-device into the index). */
int sa_hip_index_build_device(sa_hip_index* idx, const void* T_dev, uint64_t n, uint32_t max_suffix_length);
/* Build + libsais64 layout in one call (replaces libsais64 on device buffers, libsais64.c:6657-6685): as
 * sa_hip_index_build_device, and sa64_dev[i] = (int64_t)SA[i] for i in [0, n), sa64_dev a device buffer of n * 8 bytes
 * on the index's device.  The widening is not a pass of its own here: on the narrow-record plan the last pass of the
 * sort stores every suffix index as u32 (the index's own array) and as int64 (sa64_dev), and the few slots refined
 * afterwards are patched; other plans end with the widening kernel.  total_ms of the build statistics covers all of it. */
int sa_hip_index_build_device64(sa_hip_index* idx, const void* T_dev, uint64_t n, uint32_t max_suffix_length, void* sa64_dev);
/* Adopt an existing suffix array (host pointers): uploads T and SA and prepares the query
 * structures; SA must be sorted by the first max_suffix_length bytes (0 = fully sorted).
 * Every entry is range-checked on the device: an array with an entry >= n is refused (-1). */
int sa_hip_index_load(sa_hip_index* idx, const uint8_t* T_host, const uint32_t* SA_host, uint64_t n,
                      uint32_t max_suffix_length);
/* Same with device pointers (multi-GPU replicas: T and SA arrive by RCCL broadcast). */
int sa_hip_index_load_device(sa_hip_index* idx, const void* T_dev, const void* SA_dev, uint64_t n,
                             uint32_t max_suffix_length);

/* Replicas without any rebuilding (no counterpart in the reference).  The query structures of a built
 * index -- text, suffix array, sorted key array, bucket directory -- are plain device buffers; a replica reserves
 * buffers of the same layout, the caller fills them (RCCL broadcast straight into them: sa_hip_comm_replicate_index,
 * or torch.distributed over the same pointers), and commit makes the replica searchable after range-checking the
 * suffix array and the directory's ends on the device.  Nothing is gathered, sorted or searched on the replica. */
typedef struct sa_hip_replica_layout {
    uint64_t n;
    uint32_t max_suffix_length;
    uint32_t key_bytes;          /* 0: no key array (n < 2); 4: u32 narrow keys; 8: u64 keys                */
    uint32_t bits_per_symbol;    /* of the packed keys                                                      */
    uint32_t initial_chars;
    uint32_t dir_bits;
    int32_t  lo_shift;           /* narrow keys: key = (bucket << 56) | (narrow << lo_shift)                */
    uint64_t dir_entries;        /* 2^dir_bits + 1                                                          */
    uint16_t code[256];          /* alphabet compaction of the text                                         */
    uint64_t freq[256];
} sa_hip_replica_layout;
typedef struct sa_hip_replica_buffers {
    void* text; void* sa; void* keys; void* dir;          /* device pointers on the index's device        */
    uint64_t text_bytes, sa_bytes, keys_bytes, dir_bytes;  /* what has to travel                            */
} sa_hip_replica_buffers;
/* Layout and buffers of a built (or loaded) index: the source of a replication. */
int sa_hip_index_replica_layout(sa_hip_index* idx, sa_hip_replica_layout* out);
int sa_hip_index_replica_buffers(sa_hip_index* idx, sa_hip_replica_buffers* out);
/* Destination: allocate buffers for `layout` (n <= the handle's capacity) and return where to receive.  The index has
 * no searchable state until sa_hip_index_replica_commit. */
int sa_hip_index_replica_reserve(sa_hip_index* idx, const sa_hip_replica_layout* layout, sa_hip_replica_buffers* out);
int sa_hip_index_replica_commit(sa_hip_index* idx);

/* Multi-GPU lifecycle without PyTorch (the reference has no distributed code): one process
 * per GPU, RCCL over xGMI, loaded at run time (no link-time dependency; a process that already holds a librccl.so --
 * PyTorch ships one -- gets that copy).  Rank 0 calls sa_hip_comm_unique_id and ships the 128 bytes to the other ranks
 * by whatever it has (MPI, a file, a socket: RCCL's own bootstrap contract); every rank then calls sa_hip_comm_create.
 *   sa_hip_comm_replicate_index  root: a built index; other ranks: an empty handle of sufficient capacity, searchable
 *                                afterwards -- one broadcast per buffer (text, SA, key array, directory) straight into
 *                                reserved buffers, nothing rebuilt (sa_hip_index_replica_*); *bytes_out = bytes moved
 *   sa_hip_comm_allgather_ranges every rank's pairs_per_rank (first, last) pairs -> recv_dev[nranks][pairs_per_rank], on the
 *                                index's own stream (ordered after the search that produced them; asynchronous until
 *                                sa_hip_index_sync)
 * torch.distributed drives the same replica entry points in suffixarray_amd/distributed.py. */
#define SA_HIP_COMM_ID_BYTES 128
typedef struct sa_hip_comm sa_hip_comm;
int sa_hip_comm_unique_id(void* id128);
int sa_hip_comm_create(sa_hip_comm** out, const void* id128, int nranks, int rank, int device);
void sa_hip_comm_destroy(sa_hip_comm* comm);
int sa_hip_comm_rank(const sa_hip_comm* comm);
int sa_hip_comm_size(const sa_hip_comm* comm);
int sa_hip_comm_replicate_index(sa_hip_comm* comm, sa_hip_index* idx, int root, uint64_t* bytes_out);
int sa_hip_comm_allgather_ranges(sa_hip_comm* comm, sa_hip_index* idx, const void* send_dev, uint64_t pairs_per_rank, void* recv_dev);

uint64_t sa_hip_index_n(const sa_hip_index* idx);
uint32_t sa_hip_index_max_suffix_length(const sa_hip_index* idx);
/* Device pointers owned by the index: text (n bytes + zero padding) and SA (uint32[n]). */
const void* sa_hip_index_text_dev(const sa_hip_index* idx);
const void* sa_hip_index_sa_dev(const sa_hip_index* idx);
/* The index's HIP stream (hipStream_t as void*), for event timing by the caller. */
void* sa_hip_index_stream(const sa_hip_index* idx);

/* Copy the suffix array to the host: uint32[n]; int32[n] (libsais layout); int64[n] (libsais64
 * layout, widened on the device). */
int sa_hip_index_get_sa_u32(sa_hip_index* idx, uint32_t* out_host);
int sa_hip_index_get_sa_i64(sa_hip_index* idx, int64_t* out_host);
/* The suffix array in libsais64 layout, device to device: out_dev[i] = (int64_t)SA[i] for i in [0, n), out_dev a
 * device buffer of n * 8 bytes on the index's device (libsais64.c:6248-6259 widens in place on the CPU; this is
 * the same pass as one kernel, 12 bytes of HBM traffic per entry).  Asynchronous on the index's stream; its
 * HIP-event time is reported as sa_hip_build_stats.widen_ms. */
int sa_hip_index_widen_device(sa_hip_index* idx, void* out_dev);
/* 256-bin byte histogram of the indexed text (libsais `freq`). */
int sa_hip_index_get_freq(sa_hip_index* idx, uint64_t* freq256);

/* Batched get_substring_positions (engine.c:869-918 per element).  Pattern i is
 * patterns[offsets[i] .. offsets[i+1]); compare length c = min(len, max_suffix_length) when
 * the index is truncated, len otherwise.  out[i] = {first,last} inclusive SA range;
 * {UINT32_MAX,UINT32_MAX} when every suffix is smaller; miss -> first = lower bound,
 * last = first - 1 (mod 2^32).  An empty pattern matches every suffix: {0, n-1}.
 * Host pointers; H2D/D2H copies are issued on the index's stream. */
int sa_hip_query_batch(sa_hip_index* idx, const uint8_t* patterns, const uint64_t* offsets, uint64_t Q,
                       sa_hip_pair_u32* out);
/* Same with every buffer in device memory (no copies; asynchronous on the index's stream
 * until sa_hip_index_sync).  patterns_dev must stay readable for 8 bytes past offsets[Q]: a pattern's last
 * partial 8-byte word is loaded whole and masked (the host-pointer form pads its staging copy itself). */
int sa_hip_query_batch_device(sa_hip_index* idx, const void* patterns_dev, const void* offsets_dev,
                              uint64_t Q, void* out_dev);
/* The same for Q patterns of ONE length, packed back to back (pattern i = patterns_dev[i * pattern_len ..)): no offsets
 * array -- two 8-byte loads per query less (a batch is bound by the number of memory requests).  The
 * padding rule of sa_hip_query_batch_device applies (8 readable bytes past the last pattern). */
int sa_hip_query_batch_device_fixed(sa_hip_index* idx, const void* patterns_dev, uint64_t pattern_len, uint64_t Q, void* out_dev);
/* Copy up to `cap` suffix positions SA[first .. first+count) to the host (hit materialisation). */
int sa_hip_index_get_sa_range(sa_hip_index* idx, uint64_t first, uint64_t count, uint32_t* out_host);
/* ONE query with its first hits, the latency path of record retrieval (get_matching_records, engine.c:1167-1215,
 * called per query by pyx:209-267): *range as in sa_hip_query_batch, hits[0 .. *nhits) = SA[first .. first + *nhits),
 * *nhits = min(number of hits, max_hits, 4096).  Pattern, range and hits travel through one pinned host block that
 * the kernels read and write directly: two small launches and one synchronisation, no copy calls. */
int sa_hip_index_query_hits(sa_hip_index* idx, const uint8_t* pattern, uint64_t len, uint32_t max_hits,
                            sa_hip_pair_u32* range, uint32_t* hits, uint32_t* nhits);

/* Second-level keys (round 4; csrc/sa_query.hpp): for the SA slots that share their key (first k0 characters) with a neighbour,
 * the next floor(64 / b) characters, packed like the key -- 8 n bytes, one gather over those slots.  A pattern longer than the
 * key then finds its bounds inside a key group by a binary search over 8-byte keys instead of text comparisons (two dependent
 * random reads per step).  Wide-key indexes only (word / name / DNA text; near-random text has no such groups).
 *   mode 1 (the default of every handle): the first batch of >= 32768 patterns builds them on its way;
 *   mode 2: build them now;   mode 0: drop them and never build them (the text search stays).
 * Returns 1 when the index has them afterwards, 0 when not (narrow keys, no memory, mode 0), < 0 on errors.  The results of
 * every query are the same with or without; a rebuild / load / replica commit drops them. */
int sa_hip_index_deep_keys(sa_hip_index* idx, int mode);
int sa_hip_index_sync(sa_hip_index* idx);

/* On-device check that the index's SA is the suffix array of its text (truncated indexes: that
 * it is a permutation ordered by the first max_suffix_length bytes, ties in text order).
 * *violations = 0 means verified.  O(n) device work, 4n bytes of device scratch; the O(n)
 * "sufcheck" that makes bit-exactness testable at n = 1e9 without a CPU oracle run. */
int sa_hip_index_verify(sa_hip_index* idx, uint64_t* violations);

/* ---- (5) record retrieval: hits -> rows (engine.c:920-999, 1168-1215, 1326-1390; bound at pyx:87-101) -------------- */

/* Row table of the indexed text: row r (a document, or one CSV field) = text[row_text_starts[r], row_text_starts[r+1]);
 * row_text_starts[0] must be 0 and the offsets ascend.  The table is copied (under the handle's lock, like every
 * reader of it). */
int sa_hip_index_set_rows(sa_hip_index* idx, const uint64_t* row_text_starts, uint64_t num_rows);
/* ONE query -> the distinct rows that contain the pattern, in SA order of their first hit, at most k of them
 * (row_ids[0 .. *num_rows)); *range (may be NULL) as in sa_hip_query_batch.  The reference returns one record per HIT
 * (engine.c:1364-1388); one per ROW is this library's decision. */
int sa_hip_index_query_rows(sa_hip_index* idx, const uint8_t* pattern, uint64_t len, uint32_t k, uint64_t* row_ids,
                            uint32_t* num_rows, sa_hip_pair_u32* range);
/* The batched form (new surface; per element it is sa_hip_index_query_rows): pattern i = patterns[offsets[i] ..
 * offsets[i+1]); row_ids[i * k .. i * k + counts[i]) = its distinct rows, ranges[i] (may be NULL) its SA range.  ONE search
 * launch finds every range and ONE more launch maps every hit of every range to its row (binary search over the row
 * table in HBM) and de-duplicates per query in LDS (k <= 4096; larger k is served per query on the host): no per-query
 * synchronisation, one copy back.  Replaces the per-query loop of suffix_array.pyx:221-247 over engine.c:1364-1388.
 * In a batch of 4096 queries or more, ranges of at most 4 hits are answered by one lane each, longer ones by one wave each
 * (k <= 64) or one workgroup each; when Q * k row ids exceed 32 MiB the host legs go through the process's ring of pinned slabs (512 MiB, shared with
 * the sa_hip_libsais* wrappers, given back by sa_hip_release_workspace) and the ids are widened into row_ids by worker threads.
 * row_ids, counts and ranges are the caller's arrays (entries of row_ids beyond counts[i] are left untouched): keep them from
 * call to call. */
int sa_hip_index_query_rows_batch(sa_hip_index* idx, const uint8_t* patterns, const uint64_t* offsets, uint64_t Q, uint32_t k,
                                  uint64_t* row_ids, uint32_t* counts, sa_hip_pair_u32* ranges);
/* The same for a range that a batched query has already found (sa_hip_query_batch: one launch for all the ranges,
 * then the rows per range). */
int sa_hip_index_rows_for_range(sa_hip_index* idx, sa_hip_pair_u32 range, uint32_t k, uint64_t* row_ids, uint32_t* num_rows);
/* Copy the indexed text (n bytes) back to the host (persistence: the CSV-mode text is the extracted column). */
int sa_hip_index_get_text(sa_hip_index* idx, uint8_t* out_host);

/* CSV-mode index: replaces SuffixArrayIndex + construct_truncated_suffix_array_from_csv_partitioned_mmap_full
 * (engine.h:163-172, engine.c:1454-1482 -> 461-654): extracts `search_column` of an RFC-4180 file (lower-cased, one
 * '\n' after every field, header row not indexed, 64-bit file offsets), builds the device index over it with
 * max_suffix_length and keeps the row tables + a read-only mapping of the file.  One index per file: the 2 GiB
 * partitioning of the reference (engine.c:1437) is not reproduced (288 GB of HBM; the column must stay below 2^32 - 2
 * bytes). */
typedef struct sa_hip_csv_index sa_hip_csv_index;
int sa_hip_csv_index_create(sa_hip_csv_index** out, const char* csv_file, const char* search_column, uint32_t max_suffix_length,
                            int device);
/* Re-open a saved CSV-mode index without parsing or building (persistence; the reference's
 * read_suffix_array is declared but never defined, engine.h:141): adopts text + SA + row tables; column_names =
 * num_columns NUL-terminated names back to back. */
int sa_hip_csv_index_adopt(sa_hip_csv_index** out, const char* csv_file, const uint8_t* text, const uint32_t* SA, uint64_t n,
                           const uint64_t* row_text_starts, const uint64_t* row_file_offsets, uint64_t num_rows,
                           const char* column_names, uint32_t num_columns, uint32_t column_index, uint32_t max_suffix_length,
                           int device);
void sa_hip_csv_index_destroy(sa_hip_csv_index* c);
/* The same rows without the copies: row_ptrs[i] points INTO the index's read-only mapping of the CSV file (row_lens[i] bytes, line
 * terminator excluded, NOT NUL-terminated), valid until the index is destroyed; at most k rows, *num_matches = how many.  New
 * surface (the reference's interface mallocs every record, engine.c:1382): what a binding uses when it builds its own objects
 * from the bytes anyway. */
int sa_hip_get_matching_row_spans_file(sa_hip_csv_index* c, const char* substring, uint32_t k, const char** row_ptrs,
                                       uint32_t* row_lens, uint32_t* num_matches);
/* A column of more than `partition_bytes` (0 or > 2^32 - 2: 2^32 - 2) bytes as several independent indexes of WHOLE rows over
 * the same file: the reference's partitions (engine.c:1437-1481 cuts the FILE every 2 GiB; suffix_array.pyx:221-247 answers
 * from the partitions one after the other).  *out_parts: malloc'ed array of *num_parts handles (>= 1; each is destroyed with
 * sa_hip_csv_index_destroy, the array with sa_hip_csv_index_free_parts); every sa_hip_csv_index entry point works on a part.
 * The file is parsed once. */
int sa_hip_csv_index_create_partitioned(sa_hip_csv_index*** out_parts, uint32_t* num_parts, const char* csv_file, const char* search_column,
                                        uint32_t max_suffix_length, int device, uint64_t partition_bytes);
void sa_hip_csv_index_free_parts(sa_hip_csv_index** parts);
/* The device index underneath (batched queries, statistics, verification); owned by the CSV index. */
sa_hip_index* sa_hip_csv_index_handle(sa_hip_csv_index* c);
uint64_t sa_hip_csv_index_num_rows(const sa_hip_csv_index* c);
uint32_t sa_hip_csv_index_num_columns(const sa_hip_csv_index* c);
uint32_t sa_hip_csv_index_column_index(const sa_hip_csv_index* c);
const char* sa_hip_csv_index_column_name(const sa_hip_csv_index* c, uint32_t i);
/* Borrowed views of the row tables: row_text_starts[num_rows], row_file_offsets[num_rows + 1]. */
int sa_hip_csv_index_row_tables(const sa_hip_csv_index* c, const uint64_t** row_text_starts, const uint64_t** row_file_offsets);

/* Rows of the file by id, as malloc'ed NUL-terminated strings without the line terminator (records[0 .. n); ownership
 * as in sa_hip_get_matching_records_file). */
int sa_hip_csv_index_copy_rows(sa_hip_csv_index* c, const uint64_t* row_ids, uint32_t n, char** records);

/* replaces get_substring_positions_file (engine.h:235-239, engine.c:920-999): the search of CSV mode.  The reference
 * reads the text per probe from the file (fseek + fread + tolower); here it is the batched kernel with Q = 1 over the
 * extracted column in HBM.  Result conventions of THAT function: a hit -> inclusive range {first, last} over the
 * index's suffix array; ANY miss -> {UINT32_MAX, UINT32_MAX} (engine.c:962-965).  `substring` is compared as given
 * (the reference's caller lower-cases it, pyx:228). */
sa_hip_pair_u32 sa_hip_get_substring_positions_file(sa_hip_csv_index* c, const char* substring);
/* replaces get_matching_records_file (engine.h:257-264, engine.c:1326-1390; bound at pyx:94-101, called at pyx:224-232):
 * appends the rows that contain `substring` to matching_records[*num_matches ...] until *num_matches == k.  Ownership
 * as in the reference: every row is a malloc'ed NUL-terminated string (engine.c:1382), the caller frees each one
 * (pyx:262-265; or sa_hip_free_records).  Differences by decision: a row is returned once however
 * often it contains the pattern, rows come back whole (engine.c:1314 drops the last character), a miss appends
 * nothing.  Returns 0 or a negative SA_HIP_* code (the reference returns void and exit()s). */
int sa_hip_get_matching_records_file(sa_hip_csv_index* c, const char* substring, uint32_t k, char** matching_records,
                                     uint32_t* num_matches);
/* replaces get_matching_records (engine.h:249-255, engine.c:1168-1215; bound at pyx:87-93): host text + host SA, the
 * records are the lines of `str` ('\n'-separated documents) that contain the pattern; returns their number (<= k).
 * Like sa_hip_get_substring_positions this uploads text and SA for ONE query -- O(n) per call: a compatibility shim,
 * not the fast path (use a handle + sa_hip_index_set_rows + sa_hip_index_query_rows). */
uint32_t sa_hip_get_matching_records(const char* str, const sa_hip_SuffixArray_struct* sa, const char* substring, uint32_t k,
                                     char** matching_records);
/* free() every row of a result table (the table itself belongs to the caller). */
void sa_hip_free_records(char** records, uint32_t n);

/* replaces init_suffix_array_byte_idxs / free_suffix_array (engine.h:133-140, engine.c:326-349; pyx:68-74): malloc /
 * free of the caller-side uint32 suffix_array[n] of a SuffixArray_struct for the engine-compatible calls of (2) and
 * (3).  is_quoted_bitflag is left NULL: the row tables of sa_hip_csv_index replace the reference's per-character
 * quoted bits (engine.c:637-645), which only serve its newline seeks. */
int sa_hip_init_suffix_array_byte_idxs(sa_hip_SuffixArray_struct* sa, uint32_t max_suffix_length, uint64_t global_byte_start_idx,
                                       uint64_t global_byte_end_idx, uint32_t n);
void sa_hip_free_suffix_array(sa_hip_SuffixArray_struct* sa);

/* replaces write_suffix_array (engine.h:142-146, engine.c:1112-1138) and read_suffix_array (declared at engine.h:141, never
 * defined in the reference): the reference's own file layout {u64 global_byte_start_idx, u64 global_byte_end_idx,
 * u32 max_suffix_length, u32 n, u32 suffix_array[n]}, so that an index file of either side can be read by the other.
 * is_quoted_filename (may be NULL) receives an empty bit buffer {u32 capacity = 0} (engine.c:1098-1101): this library
 * keeps row tables instead of per-character quoted bits.  The reader mallocs suffix_array (sa_hip_free_suffix_array) and
 * refuses entries >= n.  Note for CSV mode: the reference stores FILE byte offsets in the array (engine.c:648-651), this
 * library positions of the extracted column; SuffixArray.save / load of the Python class keep the row tables beside it. */
int sa_hip_write_suffix_array(const sa_hip_SuffixArray_struct* sa, const char* sa_filename, const char* is_quoted_filename);
int sa_hip_read_suffix_array(sa_hip_SuffixArray_struct* sa, const char* sa_filename);

/* ---- instrumentation ---------------------------------------------------------------------- */

/* Per-build statistics of the last build on this handle (roofline accounting). */
typedef struct sa_hip_build_stats {
    uint64_t n;
    uint32_t sigma;              /* distinct byte values                                   */
    uint32_t bits_per_symbol;    /* b: code width after alphabet compaction                */
    uint32_t initial_chars;      /* K0: characters packed into the initial 64-bit key      */
    uint32_t rounds;             /* refinement rounds after the initial sort               */
    uint32_t chunk_rounds;
    uint32_t doubling_rounds;
    uint32_t final_depth;        /* h when the active set became empty                     */
    uint32_t radix_passes;       /* onesweep launches over all sorts                       */
    uint64_t radix_records;      /* sum over passes of records moved                       */
    uint64_t radix_bytes;        /* algorithmic bytes of those passes (read + written)    */
    uint64_t active_total;       /* sum over rounds of active-set sizes                    */
    uint64_t tiny_resolved;      /* suffixes ordered by the tiny-group finisher            */
    double   radix_ms;           /* HIP-event time of all onesweep launches                */
    double   total_ms;           /* HIP-event time of the whole device build               */
    /* the sort passes by kernel: [0] radix_onesweep_kernel<512> (u64 key + u32 value in and out),
     * [1] top digit of a narrow sort (u32 key + u32 value out): radix_onesweep_kernel<512,0,true> from u64
     *     keys, or text_top_pass_kernel<512> from the text (text_top_pass),
     * [2] seg_onesweep_kernel<512,24,false,true> (u32 key + u32 value in and out),
     * [3] seg_onesweep_kernel<512,24,true,true>: last narrow pass (u32 key + u32 value in; out: u32 key + u32 value when
     *     the index keeps the narrow keys (narrow_k), else rebuilt u64 key + u32 value) */
    double   pass_ms[4];
    uint64_t pass_bytes[4];      /* algorithmic bytes (read + written)                      */
    uint32_t pass_launches[4];
    uint32_t text_top_pass;      /* 1: kernel [1] was text_top_pass_kernel<512> (keys assembled from the text) */
    uint32_t narrow_k;           /* 1: the index keeps u32 narrow keys + 257 bucket bounds as its query key array   */
    double   widen_ms;           /* HIP-event time of the last sa_hip_index_widen_device after this build (0: none)  */
    uint64_t finisher_records;   /* records of groups that fit a tile, summed over the runs of the in-LDS group finisher */
    uint64_t finisher_resolved;  /* suffixes it ordered finally (they never see a global refinement round)            */
    uint32_t finisher_runs;
    uint32_t widen_fused;        /* 1: the int64 copy of sa_hip_index_build_device64 came out of the sort's last pass       */
    uint32_t narrow48;           /* 1: keys of 41..56 bits sorted as 10-byte records (u32 + u16 key parts + u32 index): kernels [1] =
                                  *    text_top_pass_kernel<512, true>, [2] / [3] = seg48_onesweep_kernel                           */
    uint32_t lite_flags;         /* 1: the first flags pass wrote no flag array (near-random text: active records staged per tile);
                                  * 2: there was no such pass at all -- the local pass of the three-pass plan did its work per sub-bucket */
    uint64_t period_resolved;    /* suffixes ordered by the periodic-run shortcut (long repeats: period_finish.hpp)              */
    uint32_t split_plan;         /* > 0 (= rb, the key bits of the split): the narrow sort ran as THREE passes over the records (radix_split.hpp): kernels [2] = seg_split_kernel<512,24>
                                  *    (one launch: the records of a bucket grouped by their next rb <= 10 key bits), [3] = local_finish_kernel (one
                                  *    launch: every group ordered completely in LDS, suffixes out as u32 and, for a 64-bit build, int64)        */
    uint32_t split_max;          /* largest group at the level taken (declined: at the finest level, > 8192); 0: the plan was not considered      */
} sa_hip_build_stats;
int sa_hip_index_build_stats(const sa_hip_index* idx, sa_hip_build_stats* out);

/* Statistics of the search launches on this handle.  kernel_ms: the last launch; kernel_ms_sum / launches: every launch
 * since the previous call of sa_hip_index_query_stats (a pipelined step issues several; the library keeps HIP events
 * for the last 32 launches and resolves them here -- the call waits for them). */
typedef struct sa_hip_query_stats {
    uint64_t q;                  /* patterns of the last launch                            */
    double   kernel_ms;          /* HIP-event time of the last search kernel               */
    double   kernel_ms_sum;      /* ... of all launches since the previous call            */
    uint32_t launches;
    uint32_t pad_;
} sa_hip_query_stats;
int sa_hip_index_query_stats(const sa_hip_index* idx, sa_hip_query_stats* out);

/* ---- host-side CSV column extractor (reference engine.c:26-96, 461-654) ----------- */

/* One column of an RFC-4180 CSV file, prepared for sa_hip_index_build: the fields lower-cased (ASCII)
 * with a '\n' after each, the offset of every row's field in that text and the byte offset of every
 * row in the file (num_rows + 1 entries: the last one is the end of the last row).  The header row
 * is not indexed.  All arrays are malloc'ed by the callee; release with sa_hip_csv_free. */
typedef struct sa_hip_csv_column {
    uint8_t*  text;
    uint64_t  text_len;
    uint64_t* row_text_starts;
    uint64_t* row_file_offsets;
    uint64_t  num_rows;
    char*     column_names;      /* num_columns NUL-terminated names, back to back */
    uint32_t  num_columns;
    uint32_t  column_index;
} sa_hip_csv_column;
int sa_hip_csv_extract_column(const char* path, const char* column, sa_hip_csv_column* out);
void sa_hip_csv_free(sa_hip_csv_column* col);
/* Synthetic `id,company_name,country` CSV of BASELINE config 5. */
int sa_hip_synth_csv(const char* path, uint64_t rows, uint64_t seed);

/* Stable LSD radix sort of n (u64 key, u32 value) records by key bits [begin_bit, end_bit) on
 * `device` (host pointers, sorted in place).  The device sort underneath every build, exported
 * so that it can be tested and profiled on its own.  values == NULL: values are the record
 * positions 0..n-1 and are not returned. */
int sa_hip_sort_pairs(uint64_t* keys, uint32_t* values, uint64_t n, int begin_bit, int end_bit, int device);

/* Synthetic text D1 `uniform27` of SURVEY.md 8(d): xorshift64 stream, 26 letters + '\n'. */
void sa_hip_synth_uniform27(uint8_t* out, uint64_t n, uint64_t seed);

const char* sa_hip_last_error(void);
const char* sa_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SA_HIP_H */
