// sa_capi.hip -- the C ABI of libsa_hip.so (include/sa_hip.h).  Single translation unit for
// gfx950: hipcc --offload-arch=gfx950 -shared -fPIC.  No CPU fallback: every entry point needs
// a HIP device and fails with SA_HIP_EHIP when there is none.
#include <exception>
#include <mutex>
#include <new>
#include <vector>

#include "common.hpp"
#include "radix_sort.hpp"
#include "sa_build.hpp"
#include "sa_query.hpp"
#include "big_build.hpp"
#include "csv_ingest.hpp"
#include "records.hpp"
#include "host_io.hpp"
#include "rows_device.hpp"
#include "host_index.hpp"
#include "comm_rccl.hpp"
#include <memory>
#include <chrono>

using namespace sa;

struct sa_hip_index {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    Builder b;
    bool has_index = false;
    // query staging (host-pointer API)
    DevBuf q_pat, q_off, q_out;
    DevBuf widen;
    u8* qh_host = nullptr;   // pinned, device-mapped block of sa_hip_index_query_hits (QH_BYTES)
    u8* qh_dev = nullptr;    // the same block as the device sees it
    // HIP events of the last QRING search launches: a pipelined caller (distributed.py: search chunk k+1 while chunk k is
    // gathered) asks for the statistics once per step, and every launch since the previous call is resolved then
    static constexpr int QRING = 32;
    hipEvent_t q_ev[QRING][2] = {};
    int q_head = 0, q_pending = 0;
    double q_sum_ms = 0.0;       // resolved launches since the last sa_hip_index_query_stats
    double q_last_ms = 0.0;
    u32 q_launches = 0;
    hipEvent_t w_begin = nullptr, w_end = nullptr;   // sa_hip_index_widen_device
    double widen_ms = 0.0;                           // < 0: recorded, not yet resolved
    sa_hip_query_stats qstats{};
    HostU64Array row_starts;   // sa_hip_index_set_rows: offset of every row (document, CSV field) in the indexed text
    DevBuf rows_dev;               // the same table in HBM (rows_device.hpp)
    DevBuf rows_coarse;            // every 256th entry of it (stays cached)
    u64 rows_coarse_n = 0;
    DevBuf r_rows, r_counts;       // results of the rows kernel (batched form)
    DevBuf r_pending;              // ... and the list of the queries its lane kernel leaves to the workgroup form (+ its length)
    std::unique_ptr<HostIndex> host;   // set: the opt-in no-GPU path of config 1 (host_index.hpp); nothing below touches HIP then
    bool receiving = false;        // sa_hip_index_replica_reserve .. _commit: the buffers are being filled by the caller
    bool k2_auto = true;           // sa_hip_index_deep_keys: large batches build the second-level keys on their way
    DevBuf qc_hist, qc_off, qc_part, qc_tmp;   // clustering of a large batch over a wide-key index (sa_query.hpp: qcluster_*)
    sa_hip_replica_layout pending{};
};

// multi-GPU lifecycle (comm_rccl.hpp): one communicator per process / GPU
struct sa_hip_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;   // replication runs here; the range gather runs on the searching index's stream
};

// CSV-mode index (SuffixArrayIndex of the reference, engine.h:163-172): the device index over one column + the row
// tables + the memory-mapped file the rows are copied out of
struct sa_hip_csv_index {
    sa_hip_index* idx = nullptr;
    HostU64Array row_file_offsets;   // num_rows + 1
    std::vector<std::string> columns;
    u32 column_index = 0;
    std::string path;
    const u8* map = nullptr;
    u64 map_len = 0;
};

namespace {

int set_device(int device) {
    SA_HIP_CHECK(hipSetDevice(device));
    return 0;
}

// resolves the `count` oldest pending launches of the event ring (blocks until they have finished)
int resolve_query_events(sa_hip_index* idx, int count) {
    for (; count > 0 && idx->q_pending > 0; --count) {
        const int slot = (idx->q_head - idx->q_pending + 2 * sa_hip_index::QRING) % sa_hip_index::QRING;
        SA_HIP_CHECK(hipEventSynchronize(idx->q_ev[slot][1]));
        float ms = 0.f;
        SA_HIP_CHECK(hipEventElapsedTime(&ms, idx->q_ev[slot][0], idx->q_ev[slot][1]));
        idx->q_sum_ms += ms;
        idx->q_last_ms = ms;
        --idx->q_pending;
    }
    return 0;
}

// second-level keys of a wide-key index (sa_query.hpp: k2_build_kernel), built once per index state; the caller holds idx->mu
// and has set the device.  Without memory for them (8 n bytes) the text search stays: not an error.
constexpr u64 K2_AUTO_BATCH = 32768;   // a batch of at least this many patterns builds them on its way (one gather over the tied slots)
int ensure_k2(sa_hip_index* idx) {
    Builder& b = idx->b;
    if (b.k2_ready || !b.qkeys || b.n < 2 || !idx->has_index) return 0;
    static const bool off = [] { const char* e = diag_env("SA_HIP_K2"); return e && atoi(e) == 0; }();
    if (off) return 0;
    int k2n = 64 / (b.q_b > 0 ? b.q_b : 1);
    if (k2n > 16) k2n = 16;
    if (b.qkeys2.ensure((size_t)b.n * 8 + 64)) return 0;
    if (b.qskeys.ensure(((size_t)b.n / SKEY_STRIDE + 2) * 8 + 64)) return 0;
    hipLaunchKernelGGL(skeys_kernel, dim3(stream_grid(b.n / SKEY_STRIDE + 1, 256)), dim3(256), 0, idx->stream, b.qkeys, b.n, b.qskeys.as<u64>());
    hipLaunchKernelGGL(k2_build_kernel, dim3(stream_grid(b.n, 256)), dim3(256), 0, idx->stream, b.qkeys, (const u32*)b.sa, (const u8*)b.text.as<u8>(), b.n,
                       b.qmap, b.q_b, b.q_k0, k2n, b.max_suffix_length, b.qkeys2.as<u64>());
    SA_HIP_CHECK(hipGetLastError());
    b.q_k2n = k2n;
    b.k2_ready = true;
    return 0;
}

QueryArgs query_args(sa_hip_index* idx, const u8* pat_dev, const u64* off_dev, u64 Q, sa_hip_pair_u32* out_dev, u64 fixed_len) {
    QueryArgs a;
    a.keys2 = idx->b.k2_ready ? idx->b.qkeys2.as<u64>() : nullptr;
    a.skeys = idx->b.k2_ready ? idx->b.qskeys.as<u64>() : nullptr;
    a.perm = nullptr;
    a.k2n = idx->b.q_k2n;
    a.fixed_len = fixed_len;
    a.text = idx->b.text.as<u8>();
    a.sa = idx->b.sa;
    a.n = idx->b.n;
    a.max_suffix_length = idx->b.max_suffix_length;
    a.patterns = pat_dev;
    a.offsets = off_dev;
    a.q = Q;
    a.out = out_dev;
    a.keys = idx->b.qkeys;
    a.keys32 = idx->b.qkeys32;
    a.lo_shift = idx->b.q_lo_shift;
    a.sector_search = idx->b.sector_search;
    a.dir = idx->b.qdir.as<u32>();
    a.b = idx->b.q_b; a.k0 = idx->b.q_k0; a.dbits = idx->b.q_dbits;
    return a;
}

int launch_query(sa_hip_index* idx, const u8* pat_dev, const u64* off_dev, u64 Q, sa_hip_pair_u32* out_dev, u64 fixed_len = 0) {
    if (idx->q_pending == sa_hip_index::QRING) { int rc = resolve_query_events(idx, 1); if (rc) return rc; }
    hipEvent_t* ev = idx->q_ev[idx->q_head];
    if (Q >= K2_AUTO_BATCH && idx->k2_auto) { int rc = ensure_k2(idx); if (rc) return rc; }
    QueryArgs a = query_args(idx, pat_dev, off_dev, Q, out_dev, fixed_len);
    // a large batch over a wide-key index is answered in the order of its patterns' first characters (qcluster_*): the buffers
    // first (no memory: the plain order), the three small passes inside the timed region
    // (from 2^20 patterns on: its ten small launches cost ~0.11 ms -- 1e6 names 0.324 vs 0.331 ms, 8e6 names 1.75 vs 2.40 ms;
    //  SA_HIP_QCLUSTER_MIN lowers the threshold for the tests, SA_HIP_QCLUSTER=0 switches it off; both under SA_HIP_DIAG=1 only)
    u64 cluster_min = 1ull << 20;
    if (const char* e = diag_env("SA_HIP_QCLUSTER_MIN")) cluster_min = strtoull(e, nullptr, 10);
    if (const char* e = diag_env("SA_HIP_QCLUSTER")) { if (atoi(e) == 0) cluster_min = ~0ull; }
    bool cluster = Q >= cluster_min && Q >= QC_TILE && Q < 0xFFFFFFFFull && a.keys && idx->k2_auto;
    const u32 qtiles = (u32)((Q + QC_TILE - 1) / QC_TILE);
    const u64 hlen = (u64)QC_BINS * qtiles, nparts = (hlen + big::SC_TILE - 1) / big::SC_TILE;
    if (cluster && (idx->qc_hist.ensure(hlen * 4 + 64) || idx->qc_off.ensure(hlen * 8 + 64) || idx->qc_part.ensure((nparts + 1) * 8 + 64) ||
                    idx->qc_tmp.ensure((size_t)Q * 16 + 64))) cluster = false;
    SA_HIP_CHECK(hipEventRecord(ev[0], idx->stream));
    if (cluster) {
        u32* qkey = idx->qc_tmp.as<u32>();
        u32* tmp = qkey + Q;
        u32* perm0 = tmp + Q;
        u32* perm1 = perm0 + Q;
        const u32 g = stream_grid(Q, 256);
        for (int pass = 0; pass < 2; ++pass) {
            const u32* order = pass ? perm0 : nullptr;
            hipLaunchKernelGGL(qcluster_rank_kernel, dim3(qtiles), dim3(256), 0, idx->stream, a, idx->b.qmap, pass, order, qkey, tmp, idx->qc_hist.as<u32>(), qtiles);
            hipLaunchKernelGGL(big::bg_scan_reduce_kernel, dim3((u32)nparts), dim3(big::SC_BLOCK), 0, idx->stream, (const u32*)idx->qc_hist.as<u32>(), hlen,
                               idx->qc_part.as<u64>());
            hipLaunchKernelGGL(big::bg_scan_parts_kernel, dim3(1), dim3(big::SC_BLOCK), 0, idx->stream, idx->qc_part.as<u64>(), nparts);
            hipLaunchKernelGGL(big::bg_scan_apply_kernel, dim3((u32)nparts), dim3(big::SC_BLOCK), 0, idx->stream, (const u32*)idx->qc_hist.as<u32>(), hlen,
                               (const u64*)idx->qc_part.as<u64>(), idx->qc_off.as<u64>());
            hipLaunchKernelGGL(qcluster_place_kernel, dim3(g), dim3(256), 0, idx->stream, Q, (const u64*)idx->qc_off.as<u64>(), (const u32*)tmp, order, qtiles,
                               pass ? perm1 : perm0);
        }
        a.perm = perm1;
    }
    if (Q) {
        u64 g = (Q + 255) / 256;
        if (g > 256u * 16u) g = 256u * 16u;
        const dim3 grid((u32)g), block(256);
        if (a.keys32) {
            if (a.sector_search == 2) hipLaunchKernelGGL((query_kernel<true, 2>), grid, block, 0, idx->stream, a, idx->b.qmap);
            else if (a.sector_search == 1) hipLaunchKernelGGL((query_kernel<true, 1>), grid, block, 0, idx->stream, a, idx->b.qmap);
            else hipLaunchKernelGGL((query_kernel<true, 0>), grid, block, 0, idx->stream, a, idx->b.qmap);
        } else {
            if (a.sector_search == 2) hipLaunchKernelGGL((query_kernel<false, 2>), grid, block, 0, idx->stream, a, idx->b.qmap);
            else if (a.sector_search == 1) hipLaunchKernelGGL((query_kernel<false, 1>), grid, block, 0, idx->stream, a, idx->b.qmap);
            else hipLaunchKernelGGL((query_kernel<false, 0>), grid, block, 0, idx->stream, a, idx->b.qmap);
        }
    }
    SA_HIP_CHECK(hipEventRecord(ev[1], idx->stream));
    SA_HIP_CHECK(hipGetLastError());
    idx->q_head = (idx->q_head + 1) % sa_hip_index::QRING;
    ++idx->q_pending;
    ++idx->q_launches;
    idx->qstats.q = Q;   // the times are resolved lazily by sa_hip_index_query_stats
    return 0;
}

// pinned block of sa_hip_index_query_hits: [0,8) range, [8,12) nhits, [64, 64 + 4*QH_MAX_HITS) hits,
// [QH_OFF_OFFSETS, +16) offsets {0, len}, [QH_OFF_PATTERN, QH_BYTES - 64) pattern (zero padded)
constexpr u32 QH_MAX_HITS = 4096;
constexpr size_t QH_OFF_HITS = 64, QH_OFF_OFFSETS = QH_OFF_HITS + 4 * QH_MAX_HITS, QH_OFF_PATTERN = QH_OFF_OFFSETS + 64;
constexpr size_t QH_BYTES = 64 * 1024;

__global__ __launch_bounds__(256) void hits_copy_kernel(const u32* __restrict__ sa, const sa_hip_pair_u32* __restrict__ range,
                                                        u32 max_hits, u32* __restrict__ hits, u32* __restrict__ nhits) {
    const u32 first = range->first, second = range->second;
    u32 count = 0;
    if (first != 0xFFFFFFFFu && ((second - first + 1u) != 0u)) count = second - first + 1u;   // miss: second = first - 1
    if (count > max_hits) count = max_hits;
    for (u32 i = threadIdx.x; i < count; i += blockDim.x) hits[i] = sa[(u64)first + i];
    if (threadIdx.x == 0) *nhits = count;
}

// ONE query + its first hits in one launch (sa_hip_index_query_hits, sa_hip_get_substring_positions[_file]): thread 0 searches,
// the workgroup copies the hits -- launch_query + hits_copy_kernel without the launch (and the two events) in between
template <bool NARROW>
__global__ __launch_bounds__(256) void query_hits_one_kernel(QueryArgs qa, CodeMap map, const u32* __restrict__ sa, u32 max_hits,
                                                             u32* __restrict__ hits, u32* __restrict__ nhits) {
    __shared__ u16 s_map[256];
    __shared__ sa_hip_pair_u32 s_rg;
    s_map[threadIdx.x] = map.code[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        const sa_hip_pair_u32 rg = query_one<NARROW, 2>(qa, s_map, 0);
        qa.out[0] = rg;
        s_rg = rg;
    }
    __syncthreads();
    const u32 first = s_rg.first, second = s_rg.second;
    u32 count = 0;
    if (first != 0xFFFFFFFFu && ((second - first + 1u) != 0u)) count = second - first + 1u;   // miss: second = first - 1
    if (count > max_hits) count = max_hits;
    for (u32 i = threadIdx.x; i < count; i += blockDim.x) hits[i] = sa[(u64)first + i];
    if (threadIdx.x == 0) *nhits = count;
}

// one-shot helper for the libsais-/engine-compatible wrappers
struct TempIndex {
    sa_hip_index* idx = nullptr;
    ~TempIndex() { if (idx) sa_hip_index_destroy(idx); }
};

// The process-level workspace of the libsais-call-compatible wrappers (further down): a cached index handle and the ring of
// pinned slabs for both legs over PCIe.  The ring also carries the large host legs of sa_hip_index_query_rows_batch (under
// `mu`; lock order: a caller's own idx->mu first, then g_oneshot.mu -- the wrappers take g_oneshot.mu and then the mutex of
// the CACHED handle, which no caller ever holds).
struct OneShot {
    std::mutex mu;
    sa_hip_index* idx = nullptr;
    PinnedRing ring;
    sa_hip_call_breakdown last{};
} g_oneshot;

}  // namespace

extern "C" {

const char* sa_hip_last_error(void) { return last_error().c_str(); }
const char* sa_hip_version(void) { return "suffixarray_amd 0.1 (gfx950)"; }

int sa_hip_device_count(void) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) return fail(SA_HIP_EHIP, "hipGetDeviceCount", hipGetErrorString(e));
    return c;
}

int sa_hip_index_create(sa_hip_index** out, uint64_t n_max, int device) {
    if (!out) return fail(SA_HIP_EINVAL, "sa_hip_index_create: out == NULL");
    *out = nullptr;
    if (n_max > 0xFFFFFFFEull) return fail(SA_HIP_EINVAL, "sa_hip_index_create: n_max exceeds 2^32 - 2");
    int cnt = sa_hip_device_count();
    if (cnt <= 0 && host_path_allowed()) {
        // no usable HIP device and the caller opted in (SA_HIP_ALLOW_HOST=1): the small host implementation of config 1
        if (n_max > HOST_MAX_N) return fail(SA_HIP_EINVAL, "sa_hip_index_create: the host path (no HIP device) holds at most 2^24 bytes");
        sa_hip_index* h = new (std::nothrow) sa_hip_index();
        if (!h) return fail(SA_HIP_ENOMEM, "sa_hip_index_create: host allocation");
        try { h->host.reset(new HostIndex()); } catch (const std::bad_alloc&) { delete h; return fail(SA_HIP_ENOMEM, "sa_hip_index_create: host allocation"); }
        h->host->n_max = n_max;
        h->device = -1;
        *out = h;
        return 0;
    }
    if (cnt < 0) return cnt;
    if (device < 0 || device >= cnt) return fail(SA_HIP_EHIP, "sa_hip_index_create: no such HIP device");
    int rc = set_device(device);
    if (rc) return rc;
    sa_hip_index* idx = new (std::nothrow) sa_hip_index();
    if (!idx) return fail(SA_HIP_ENOMEM, "sa_hip_index_create: host allocation");
    idx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete idx; return fail(SA_HIP_EHIP, "hipStreamCreate", hipGetErrorString(e)); }
    rc = idx->b.init(n_max, idx->stream);
    if (!rc) {
        bool ok = hipEventCreate(&idx->w_begin) == hipSuccess && hipEventCreate(&idx->w_end) == hipSuccess;
        for (int i = 0; ok && i < sa_hip_index::QRING; ++i)
            ok = hipEventCreate(&idx->q_ev[i][0]) == hipSuccess && hipEventCreate(&idx->q_ev[i][1]) == hipSuccess;
        if (!ok) rc = fail(SA_HIP_EHIP, "hipEventCreate");
    }
    if (rc) { sa_hip_index_destroy(idx); return rc; }
    *out = idx;
    return 0;
}

void sa_hip_index_destroy(sa_hip_index* idx) {
    if (!idx) return;
    if (idx->host) { delete idx; return; }
    (void)hipSetDevice(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    idx->b.destroy();
    idx->q_pat.release(); idx->q_off.release(); idx->q_out.release(); idx->widen.release();
    idx->rows_dev.release(); idx->rows_coarse.release(); idx->r_rows.release(); idx->r_counts.release(); idx->r_pending.release();
    idx->qc_hist.release(); idx->qc_off.release(); idx->qc_part.release(); idx->qc_tmp.release();
    if (idx->qh_host) (void)hipHostFree(idx->qh_host);
    for (int i = 0; i < sa_hip_index::QRING; ++i)
        for (int k = 0; k < 2; ++k) if (idx->q_ev[i][k]) (void)hipEventDestroy(idx->q_ev[i][k]);
    if (idx->w_begin) (void)hipEventDestroy(idx->w_begin);
    if (idx->w_end) (void)hipEventDestroy(idx->w_end);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    delete idx;
}

int sa_hip_index_build(sa_hip_index* idx, const uint8_t* T_host, uint64_t n, uint32_t max_suffix_length) {
    if (!idx || (!T_host && n)) return fail(SA_HIP_EINVAL, "sa_hip_index_build: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) {
        HostIndex& h = *idx->host;
        if (n > h.n_max) return fail(SA_HIP_EINVAL, "sa_hip_index_build: n exceeds the index capacity");
        try { h.set_text(T_host, n); h.build(max_suffix_length); }
        catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_index_build: out of host memory"); }
        idx->has_index = true;
        return 0;
    }
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (n > idx->b.n_max) return fail(SA_HIP_EINVAL, "sa_hip_index_build: n exceeds the index capacity");
    if (n) SA_HIP_CHECK(hipMemcpyAsync(idx->b.text.p, T_host, n, hipMemcpyHostToDevice, idx->stream));
    idx->has_index = false;
    idx->widen_ms = 0.0;
    rc = idx->b.build(n, max_suffix_length);
    idx->has_index = (rc == 0);
    return rc;
}

int sa_hip_index_build_device(sa_hip_index* idx, const void* T_dev, uint64_t n, uint32_t max_suffix_length) {
    if (!idx || (!T_dev && n)) return fail(SA_HIP_EINVAL, "sa_hip_index_build_device: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_index_build_device: the host path (no HIP device) has no device buffers");
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (n > idx->b.n_max) return fail(SA_HIP_EINVAL, "sa_hip_index_build_device: n exceeds the index capacity");
    if (n && T_dev != idx->b.text.p)
        SA_HIP_CHECK(hipMemcpyAsync(idx->b.text.p, T_dev, n, hipMemcpyDeviceToDevice, idx->stream));
    idx->has_index = false;
    idx->widen_ms = 0.0;
    rc = idx->b.build(n, max_suffix_length);
    idx->has_index = (rc == 0);
    return rc;
}

int sa_hip_index_build_device64(sa_hip_index* idx, const void* T_dev, uint64_t n, uint32_t max_suffix_length, void* sa64_dev) {
    if (!idx || (!T_dev && n) || (!sa64_dev && n)) return fail(SA_HIP_EINVAL, "sa_hip_index_build_device64: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_index_build_device64: the host path (no HIP device) has no device buffers");
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (n > idx->b.n_max) return fail(SA_HIP_EINVAL, "sa_hip_index_build_device64: n exceeds the index capacity");
    if (n && T_dev != idx->b.text.p)
        SA_HIP_CHECK(hipMemcpyAsync(idx->b.text.p, T_dev, n, hipMemcpyDeviceToDevice, idx->stream));
    idx->has_index = false;
    idx->widen_ms = 0.0;
    rc = idx->b.build(n, max_suffix_length, static_cast<int64_t*>(sa64_dev));
    idx->has_index = (rc == 0);
    return rc;
}

static int load_common(sa_hip_index* idx, const void* T, const void* SA, uint64_t n, uint32_t L, hipMemcpyKind kind) {
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) {
        if (kind != hipMemcpyHostToDevice) return fail(SA_HIP_EHIP, "sa_hip_index_load_device: the host path (no HIP device) has no device buffers");
        HostIndex& h = *idx->host;
        if (n > h.n_max) return fail(SA_HIP_EINVAL, "sa_hip_index_load: n exceeds the index capacity");
        const u32* s32 = static_cast<const u32*>(SA);
        for (u64 i = 0; i < n; ++i) if (s32[i] >= n) return fail(SA_HIP_EINVAL, "sa_hip_index_load: suffix array holds entries >= n");
        try { h.set_text(static_cast<const u8*>(T), n); h.sa.assign(s32, s32 + n); }
        catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_index_load: out of host memory"); }
        h.L = L; h.ready = true;
        idx->has_index = true;
        return 0;
    }
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (n > idx->b.n_max) return fail(SA_HIP_EINVAL, "sa_hip_index_load: n exceeds the index capacity");
    Builder& b = idx->b;
    if ((rc = b.sa_own.ensure((size_t)(n ? n : 1) * 4))) return rc;
    idx->has_index = false;
    u64 bad_host = 0;
    if (n) {
        SA_HIP_CHECK(hipMemcpyAsync(b.text.p, T, n, kind, idx->stream));
        SA_HIP_CHECK(hipMemcpyAsync(b.sa_own.p, SA, (size_t)n * 4, kind, idx->stream));
        // an entry >= n would send the query kernel's text reads out of bounds: refuse the array (the copy of the
        // count rides on the synchronisation prepare_text() does anyway)
        u64* bad = reinterpret_cast<u64*>(b.small.as<u8>() + 3072);
        SA_HIP_CHECK(hipMemsetAsync(bad, 0, 8, idx->stream));
        hipLaunchKernelGGL(sa_range_check_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, idx->stream,
                           (const u32*)b.sa_own.p, n, bad);
        SA_HIP_CHECK(hipMemcpyAsync(&bad_host, bad, 8, hipMemcpyDeviceToHost, idx->stream));
    }
    CodeMap map; u32 sigma; int bits;
    if ((rc = b.prepare_text(n, map, sigma, bits))) return rc;
    if (bad_host) return fail(SA_HIP_EINVAL, "sa_hip_index_load: suffix array holds entries >= n");
    b.sa = b.sa_own.as<u32>();
    b.max_suffix_length = L;
    if ((rc = b.prepare_query_from_sa(map, bits, L))) return rc;
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    idx->has_index = true;
    return 0;
}

int sa_hip_index_load(sa_hip_index* idx, const uint8_t* T_host, const uint32_t* SA_host, uint64_t n,
                      uint32_t max_suffix_length) {
    if (!idx || ((!T_host || !SA_host) && n)) return fail(SA_HIP_EINVAL, "sa_hip_index_load: NULL argument");
    return load_common(idx, T_host, SA_host, n, max_suffix_length, hipMemcpyHostToDevice);
}

int sa_hip_index_load_device(sa_hip_index* idx, const void* T_dev, const void* SA_dev, uint64_t n,
                             uint32_t max_suffix_length) {
    if (!idx || ((!T_dev || !SA_dev) && n)) return fail(SA_HIP_EINVAL, "sa_hip_index_load_device: NULL argument");
    return load_common(idx, T_dev, SA_dev, n, max_suffix_length, hipMemcpyDeviceToDevice);
}

// ---- replicas: the query structures travel as they are -----------------------------------------------
// A replica used to receive text + SA and rebuild the key array by a random gather over the text and the directory by
// binary search (sa_hip_index_load_device: 115 ms at n = 1e9, 5x a build).  The building index already holds all of it:
// the replica reserves buffers of the same layout, the caller fills them (RCCL broadcast straight into them), commit
// range-checks the suffix array.

static void fill_layout(const sa_hip_index* idx, sa_hip_replica_layout* out) {
    const Builder& b = idx->b;
    memset(out, 0, sizeof *out);
    out->n = b.n;
    out->max_suffix_length = b.max_suffix_length;
    out->key_bytes = b.qkeys32 ? 4u : (b.qkeys ? 8u : 0u);
    out->bits_per_symbol = (uint32_t)b.q_b;
    out->initial_chars = (uint32_t)b.q_k0;
    out->dir_bits = out->key_bytes ? (uint32_t)b.q_dbits : 0u;
    out->lo_shift = b.q_lo_shift;
    out->dir_entries = out->key_bytes ? (1ull << b.q_dbits) + 1 : 0;
    memcpy(out->code, b.qmap.code, sizeof out->code);
    memcpy(out->freq, b.freq, sizeof out->freq);
}

int sa_hip_index_replica_layout(sa_hip_index* idx, sa_hip_replica_layout* out) {
    if (!idx || !out) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_layout: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_index_replica_layout: the host path (no HIP device) has no device buffers");
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_layout: no index");
    fill_layout(idx, out);
    return 0;
}

static void fill_buffers(const sa_hip_index* idx, const sa_hip_replica_layout& l, sa_hip_replica_buffers* out) {
    const Builder& b = idx->b;
    memset(out, 0, sizeof *out);
    out->text = b.text.p;                       out->text_bytes = l.n;
    out->sa = b.sa;                             out->sa_bytes = l.n * 4;
    out->keys = l.key_bytes == 4 ? (void*)b.qkeys32 : (void*)b.qkeys;
    out->keys_bytes = l.n * l.key_bytes;
    out->dir = l.key_bytes ? b.qdir.p : nullptr; out->dir_bytes = l.dir_entries * 4;
}

int sa_hip_index_replica_buffers(sa_hip_index* idx, sa_hip_replica_buffers* out) {
    if (!idx || !out) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_buffers: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_index_replica_buffers: the host path (no HIP device) has no device buffers");
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_buffers: no index");
    int rc = set_device(idx->device);
    if (rc) return rc;
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));   // whoever reads the buffers next runs on another stream
    sa_hip_replica_layout l;
    fill_layout(idx, &l);
    fill_buffers(idx, l, out);
    return 0;
}

int sa_hip_index_replica_reserve(sa_hip_index* idx, const sa_hip_replica_layout* l, sa_hip_replica_buffers* out) {
    if (!idx || !l || !out) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_reserve: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_index_replica_reserve: the host path (no HIP device) has no device buffers");
    if (l->n > idx->b.n_max) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_reserve: n exceeds the index capacity");
    if (l->key_bytes != 0 && l->key_bytes != 4 && l->key_bytes != 8) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_reserve: bad key width");
    if (l->key_bytes && (l->dir_bits < 8 || l->dir_bits > 28 || l->dir_entries != (1ull << l->dir_bits) + 1 || l->bits_per_symbol < 1 ||
                         l->bits_per_symbol > 16 || l->initial_chars < 1 || l->initial_chars * l->bits_per_symbol > 64 ||
                         l->lo_shift < 0 || l->lo_shift > 63))
        return fail(SA_HIP_EINVAL, "sa_hip_index_replica_reserve: inconsistent layout");
    int rc = set_device(idx->device);
    if (rc) return rc;
    Builder& b = idx->b;
    idx->has_index = false;
    if ((rc = b.sa_own.ensure((size_t)(l->n ? l->n : 1) * 4))) return rc;
    if (l->key_bytes) {
        if ((rc = b.keys0.ensure((size_t)l->n * l->key_bytes + 64))) return rc;
        if ((rc = b.qdir.ensure((size_t)l->dir_entries * 4))) return rc;
    }
    SA_HIP_CHECK(hipMemsetAsync(b.text.as<u8>() + l->n, 0, TEXT_PAD + 16, idx->stream));
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    b.n = l->n;
    b.sa = b.sa_own.as<u32>();
    b.qkeys = nullptr; b.qkeys32 = nullptr; b.dir_ready = false; b.k2_ready = false;
    idx->pending = *l;
    idx->receiving = true;
    memset(out, 0, sizeof *out);
    out->text = b.text.p;     out->text_bytes = l->n;
    out->sa = b.sa_own.p;     out->sa_bytes = l->n * 4;
    out->keys = l->key_bytes ? b.keys0.p : nullptr; out->keys_bytes = l->n * l->key_bytes;
    out->dir = l->key_bytes ? b.qdir.p : nullptr;   out->dir_bytes = l->dir_entries * 4;
    return 0;
}

int sa_hip_index_replica_commit(sa_hip_index* idx) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_commit: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_index_replica_commit: the host path (no HIP device) has no device buffers");
    if (!idx->receiving) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_commit: nothing reserved");
    int rc = set_device(idx->device);
    if (rc) return rc;
    Builder& b = idx->b;
    const sa_hip_replica_layout& l = idx->pending;
    idx->receiving = false;
    // what arrived is checked as far as it can send the search out of bounds: SA entries < n, the directory's ends
    u64 bad_host = 0;
    u32 dir_ends[2] = {0, (u32)l.n};
    if (l.n) {
        u64* bad = reinterpret_cast<u64*>(b.small.as<u8>() + 3072);
        SA_HIP_CHECK(hipMemsetAsync(bad, 0, 8, idx->stream));
        hipLaunchKernelGGL(sa_range_check_kernel, dim3(stream_grid(l.n, 1024)), dim3(256), 0, idx->stream, (const u32*)b.sa_own.p, l.n, bad);
        SA_HIP_CHECK(hipMemcpyAsync(&bad_host, bad, 8, hipMemcpyDeviceToHost, idx->stream));
        if (l.key_bytes) {
            SA_HIP_CHECK(hipMemcpyAsync(&dir_ends[0], b.qdir.as<u32>(), 4, hipMemcpyDeviceToHost, idx->stream));
            SA_HIP_CHECK(hipMemcpyAsync(&dir_ends[1], b.qdir.as<u32>() + (l.dir_entries - 1), 4, hipMemcpyDeviceToHost, idx->stream));
        }
        SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    }
    if (bad_host) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_commit: suffix array holds entries >= n");
    if (dir_ends[0] != 0 || dir_ends[1] != (u32)l.n) return fail(SA_HIP_EINVAL, "sa_hip_index_replica_commit: directory does not span the suffix array");
    b.max_suffix_length = l.max_suffix_length;
    memcpy(b.freq, l.freq, sizeof b.freq);
    memcpy(b.qmap.code, l.code, sizeof l.code);
    b.q_b = (int)l.bits_per_symbol; b.q_k0 = (int)l.initial_chars; b.q_dbits = (int)l.dir_bits; b.q_lo_shift = l.lo_shift;
    b.qkeys = l.key_bytes == 8 ? b.keys0.as<u64>() : nullptr;
    b.qkeys32 = l.key_bytes == 4 ? b.keys0.as<u32>() : nullptr;
    b.k2_ready = false;
    b.dir_ready = l.key_bytes != 0;
    idx->has_index = true;
    return 0;
}

// ---- multi-GPU lifecycle without PyTorch -----------------------------------------------------

int sa_hip_comm_unique_id(void* id128) {
    if (!id128) return fail(SA_HIP_EINVAL, "sa_hip_comm_unique_id: NULL argument");
    RcclApi& r = rccl_api();
    if (!r.ok) return fail(SA_HIP_EHIP, "sa_hip_comm: librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == SA_HIP_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    SA_RCCL_CHECK(r.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return 0;
}

int sa_hip_comm_create(sa_hip_comm** out, const void* id128, int nranks, int rank, int device) {
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(SA_HIP_EINVAL, "sa_hip_comm_create: invalid arguments");
    *out = nullptr;
    RcclApi& r = rccl_api();
    if (!r.ok) return fail(SA_HIP_EHIP, "sa_hip_comm: librccl.so could not be loaded");
    int rc = set_device(device);
    if (rc) return rc;
    sa_hip_comm* c = new (std::nothrow) sa_hip_comm();
    if (!c) return fail(SA_HIP_ENOMEM, "sa_hip_comm_create: host allocation");
    c->nranks = nranks; c->rank = rank; c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t nr = r.CommInitRank(&c->comm, nranks, id, rank);
    if (nr != ncclSuccess) { delete c; return fail(SA_HIP_EHIP, "ncclCommInitRank", r.GetErrorString(nr)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { (void)r.CommDestroy(c->comm); delete c; return fail(SA_HIP_EHIP, "hipStreamCreate"); }
    *out = c;
    return 0;
}

void sa_hip_comm_destroy(sa_hip_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->comm) (void)rccl_api().CommDestroy(c->comm);
    delete c;
}

int sa_hip_comm_rank(const sa_hip_comm* c) { return c ? c->rank : -1; }
int sa_hip_comm_size(const sa_hip_comm* c) { return c ? c->nranks : 0; }

int sa_hip_comm_replicate_index(sa_hip_comm* c, sa_hip_index* idx, int root, uint64_t* bytes_out) {
    if (!c || !idx || root < 0 || root >= c->nranks) return fail(SA_HIP_EINVAL, "sa_hip_comm_replicate_index: invalid arguments");
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_comm_replicate_index: the host path (no HIP device) has no device buffers");
    RcclApi& r = rccl_api();
    int rc = set_device(c->device);
    if (rc) return rc;
    // Every fallible LOCAL step comes before a collective and its outcome travels with it, so that all ranks either run all
    // four buffer broadcasts or all return: (1) the root's layout goes out together with the root's status -- a root without
    // an index still broadcasts, with the error set; (2) after every rank has reserved (n > capacity, out of memory) the
    // ranks agree on the worst status by one all-reduce.  A rank that returned early would leave its peers inside
    // ncclBroadcast for good.
    struct Wire { int32_t status; int32_t pad; sa_hip_replica_layout lay; } w;
    memset(&w, 0, sizeof w);
    int rc_local = 0;
    if (c->rank == root) { rc_local = sa_hip_index_replica_layout(idx, &w.lay); w.status = rc_local; }
    void* bounce = nullptr;
    if (hipMalloc(&bounce, sizeof w + 16) != hipSuccess) {
        // (no device memory for ~5 KB: the peers cannot be told either; they fail in the broadcast's own error path)
        return fail(SA_HIP_ENOMEM, "sa_hip_comm_replicate_index: hipMalloc of the bounce buffer");
    }
    int32_t* agree = reinterpret_cast<int32_t*>(static_cast<u8*>(bounce) + sizeof w);
    auto body = [&]() -> int {
        SA_HIP_CHECK(hipMemcpyAsync(bounce, &w, sizeof w, hipMemcpyHostToDevice, c->stream));
        SA_RCCL_CHECK(r.Broadcast(bounce, bounce, sizeof w, ncclUint8, root, c->comm, c->stream));
        SA_HIP_CHECK(hipMemcpyAsync(&w, bounce, sizeof w, hipMemcpyDeviceToHost, c->stream));
        SA_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (w.status != 0)   // the same decision on every rank: nobody enters the buffer broadcasts
            return (c->rank == root) ? rc_local : fail(w.status, "sa_hip_comm_replicate_index: the root has no index to replicate");
        sa_hip_replica_buffers b;
        memset(&b, 0, sizeof b);
        rc_local = (c->rank == root) ? sa_hip_index_replica_buffers(idx, &b) : sa_hip_index_replica_reserve(idx, &w.lay, &b);
        int32_t mine = rc_local, worst = 0;   // error codes are negative: the minimum is the worst
        SA_HIP_CHECK(hipMemcpyAsync(agree, &mine, 4, hipMemcpyHostToDevice, c->stream));
        SA_RCCL_CHECK(r.AllReduce(agree, agree, 1, ncclInt32, ncclMin, c->comm, c->stream));
        SA_HIP_CHECK(hipMemcpyAsync(&worst, agree, 4, hipMemcpyDeviceToHost, c->stream));
        SA_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (worst != 0)
            return rc_local ? rc_local : fail(worst, "sa_hip_comm_replicate_index: another rank could not provide / reserve its buffers");
        void* ptr[4] = {b.text, b.sa, b.keys, b.dir};
        const u64 len[4] = {b.text_bytes, b.sa_bytes, b.keys_bytes, b.dir_bytes};
        u64 total = 0;
        for (int i = 0; i < 4; ++i) {
            if (!len[i]) continue;
            SA_RCCL_CHECK(r.Broadcast(ptr[i], ptr[i], (size_t)len[i], ncclUint8, root, c->comm, c->stream));
            total += len[i];
        }
        SA_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (bytes_out) *bytes_out = total;
        return (c->rank == root) ? 0 : sa_hip_index_replica_commit(idx);   // local, after the last collective
    };
    rc = body();
    (void)hipFree(bounce);
    return rc;
}

int sa_hip_comm_allgather_ranges(sa_hip_comm* c, sa_hip_index* idx, const void* send_dev, uint64_t pairs_per_rank, void* recv_dev) {
    if (!c || !idx || ((!send_dev || !recv_dev) && pairs_per_rank)) return fail(SA_HIP_EINVAL, "sa_hip_comm_allgather_ranges: NULL argument");
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_comm_allgather_ranges: the host path (no HIP device) has no device buffers");
    if (pairs_per_rank == 0) return 0;
    int rc = set_device(c->device);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(idx->mu);
    // on the index's own stream: ordered after the search that produced send_dev, no host synchronisation in between
    SA_RCCL_CHECK(rccl_api().AllGather(send_dev, recv_dev, (size_t)pairs_per_rank * sizeof(sa_hip_pair_u32), ncclUint8, c->comm, idx->stream));
    return 0;
}

uint64_t sa_hip_index_n(const sa_hip_index* idx) { return idx ? (idx->host ? idx->host->n : idx->b.n) : 0; }
uint32_t sa_hip_index_max_suffix_length(const sa_hip_index* idx) { return idx ? (idx->host ? idx->host->L : idx->b.max_suffix_length) : 0; }
const void* sa_hip_index_text_dev(const sa_hip_index* idx) { return (idx && !idx->host) ? idx->b.text.p : nullptr; }
const void* sa_hip_index_sa_dev(const sa_hip_index* idx) { return (idx && idx->has_index && !idx->host) ? idx->b.sa : nullptr; }
void* sa_hip_index_stream(const sa_hip_index* idx) { return idx ? (void*)idx->stream : nullptr; }

int sa_hip_index_deep_keys(sa_hip_index* idx, int mode) {
    if (!idx || mode < 0 || mode > 2) return fail(SA_HIP_EINVAL, "sa_hip_index_deep_keys: invalid arguments");
    if (idx->host) return 0;
    std::lock_guard<std::mutex> lock(idx->mu);
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (mode == 0) {   // never: drop them (8 n bytes back), large batches do not build them
        SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
        idx->k2_auto = false;
        idx->b.k2_ready = false;
        idx->b.qkeys2.release();
        idx->b.qskeys.release();
        return 0;
    }
    idx->k2_auto = true;
    if (mode == 2) {
        if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_deep_keys: no index");
        if ((rc = ensure_k2(idx))) return rc;
        SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    }
    return idx->b.k2_ready ? 1 : 0;
}

int sa_hip_index_sync(sa_hip_index* idx) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_index_sync: NULL index");
    if (idx->host) return 0;
    int rc = set_device(idx->device);
    if (rc) return rc;
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    return 0;
}

int sa_hip_index_verify(sa_hip_index* idx, uint64_t* violations) {
    if (!idx || !violations) return fail(SA_HIP_EINVAL, "sa_hip_index_verify: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);   // has_index / n are only read under the lock (a concurrent build rewrites them)
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_verify: no index");
    if (idx->host) { *violations = idx->host->verify(); return 0; }
    int rc = set_device(idx->device);
    if (rc) return rc;
    return idx->b.verify(violations);
}

int sa_hip_index_get_sa_u32(sa_hip_index* idx, uint32_t* out_host) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_u32: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_u32: no index");
    if (!out_host && sa_hip_index_n(idx)) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_u32: NULL output");
    if (idx->host) { if (idx->host->n) memcpy(out_host, idx->host->sa.data(), (size_t)idx->host->n * 4); return 0; }
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (idx->b.n) SA_HIP_CHECK(hipMemcpyAsync(out_host, idx->b.sa, (size_t)idx->b.n * 4, hipMemcpyDeviceToHost, idx->stream));
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    return 0;
}

int sa_hip_index_get_sa_i64(sa_hip_index* idx, int64_t* out_host) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_i64: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_i64: no index");
    if (!out_host && sa_hip_index_n(idx)) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_i64: NULL output");
    if (idx->host) { for (u64 i = 0; i < idx->host->n; ++i) out_host[i] = (int64_t)idx->host->sa[i]; return 0; }
    int rc = set_device(idx->device);
    if (rc) return rc;
    const u64 n = idx->b.n;
    if (n) {
        // widen on the device in slabs (libsais64.c:6248-6259 does this on the CPU, in place)
        const u64 slab = 64ull << 20;
        if ((rc = idx->widen.ensure((size_t)(n < slab ? n : slab) * 8))) return rc;
        for (u64 o = 0; o < n; o += slab) {
            const u64 cnt = (n - o) < slab ? (n - o) : slab;
            hipLaunchKernelGGL(widen_kernel, dim3(stream_grid(cnt, 1024)), dim3(256), 0, idx->stream, idx->b.sa + o, cnt,
                               idx->widen.as<int64_t>());
            SA_HIP_CHECK(hipMemcpyAsync(out_host + o, idx->widen.p, (size_t)cnt * 8, hipMemcpyDeviceToHost, idx->stream));
            SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
        }
    }
    return 0;
}

int sa_hip_index_widen_device(sa_hip_index* idx, void* out_dev) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_index_widen_device: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_index_widen_device: the host path (no HIP device) has no device buffers");
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_widen_device: no index");
    const u64 n = idx->b.n;
    if (!out_dev && n) return fail(SA_HIP_EINVAL, "sa_hip_index_widen_device: NULL output");
    int rc = set_device(idx->device);
    if (rc) return rc;
    SA_HIP_CHECK(hipEventRecord(idx->w_begin, idx->stream));
    if (n) hipLaunchKernelGGL(widen_kernel, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, idx->stream, (const u32*)idx->b.sa, n,
                              static_cast<int64_t*>(out_dev));
    SA_HIP_CHECK(hipEventRecord(idx->w_end, idx->stream));
    SA_HIP_CHECK(hipGetLastError());
    idx->widen_ms = -1.0;   // resolved by sa_hip_index_build_stats
    return 0;
}

// (idx->mu held)
static int get_sa_range_locked(sa_hip_index* idx, uint64_t first, uint64_t count, uint32_t* out_host) {
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_range: no index");
    const u64 n_idx = idx->host ? idx->host->n : idx->b.n;
    if (first > n_idx || count > n_idx - first) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_range: out of range");
    if (!out_host && count) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_range: NULL output");
    if (idx->host) { if (count) memcpy(out_host, idx->host->sa.data() + first, (size_t)count * 4); return 0; }
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (count) SA_HIP_CHECK(hipMemcpyAsync(out_host, idx->b.sa + first, (size_t)count * 4, hipMemcpyDeviceToHost, idx->stream));
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    return 0;
}

int sa_hip_index_get_sa_range(sa_hip_index* idx, uint64_t first, uint64_t count, uint32_t* out_host) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_index_get_sa_range: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    return get_sa_range_locked(idx, first, count, out_host);
}

int sa_hip_index_get_freq(sa_hip_index* idx, uint64_t* freq256) {
    if (!idx || !freq256) return fail(SA_HIP_EINVAL, "sa_hip_index_get_freq: NULL argument");
    memcpy(freq256, idx->host ? idx->host->freq : idx->b.freq, sizeof idx->b.freq);
    return 0;
}

int sa_hip_query_batch(sa_hip_index* idx, const uint8_t* patterns, const uint64_t* offsets, uint64_t Q,
                       sa_hip_pair_u32* out) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_query_batch: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_query_batch: no index");
    if (Q == 0) return 0;
    if (!offsets || !out) return fail(SA_HIP_EINVAL, "sa_hip_query_batch: NULL argument");
    const u64 total = offsets[Q];
    if (!patterns && total) return fail(SA_HIP_EINVAL, "sa_hip_query_batch: NULL patterns");
    if (idx->host) {
        for (u64 i = 0; i < Q; ++i) out[i] = idx->host->query(patterns + offsets[i], offsets[i + 1] - offsets[i]);
        return 0;
    }
    int rc = set_device(idx->device);
    if (rc) return rc;
    // a small batch (a handful of patterns: what a caller of the single-query interfaces sends) goes through the mapped pinned block
    // -- no staged copies, one launch, one synchronisation: 55 -> ~28 us for one pattern
    constexpr u64 SMALL_Q = 448, SMALL_BYTES = 16384;
    constexpr size_t SB_OUT = 0, SB_OFF = 4096, SB_PAT = 8192;
    static_assert(SMALL_Q * 8 <= SB_OFF && (SMALL_Q + 1) * 8 <= SB_PAT - SB_OFF && SB_PAT + SMALL_BYTES + 64 <= QH_BYTES, "small-batch layout");
    if (Q <= SMALL_Q && total <= SMALL_BYTES) {
        if (!idx->qh_host) {
            SA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&idx->qh_host), QH_BYTES, hipHostMallocMapped));
            SA_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&idx->qh_dev), idx->qh_host, 0));
        }
        u8* h = idx->qh_host;
        u8* d = idx->qh_dev;
        memcpy(h + SB_OFF, offsets, (size_t)(Q + 1) * 8);
        if (total) memcpy(h + SB_PAT, patterns, (size_t)total);
        memset(h + SB_PAT + total, 0, 64);
        if ((rc = launch_query(idx, d + SB_PAT, reinterpret_cast<const u64*>(d + SB_OFF), Q, reinterpret_cast<sa_hip_pair_u32*>(d + SB_OUT)))) return rc;
        SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
        memcpy(out, h + SB_OUT, (size_t)Q * sizeof(sa_hip_pair_u32));
        return 0;
    }
    if ((rc = idx->q_pat.ensure((size_t)total + 64))) return rc;
    if ((rc = idx->q_off.ensure((size_t)(Q + 1) * 8))) return rc;
    if ((rc = idx->q_out.ensure((size_t)Q * sizeof(sa_hip_pair_u32)))) return rc;
    if (total) SA_HIP_CHECK(hipMemcpyAsync(idx->q_pat.p, patterns, total, hipMemcpyHostToDevice, idx->stream));
    SA_HIP_CHECK(hipMemsetAsync(idx->q_pat.as<u8>() + total, 0, 64, idx->stream));
    SA_HIP_CHECK(hipMemcpyAsync(idx->q_off.p, offsets, (size_t)(Q + 1) * 8, hipMemcpyHostToDevice, idx->stream));
    if ((rc = launch_query(idx, idx->q_pat.as<u8>(), idx->q_off.as<u64>(), Q, idx->q_out.as<sa_hip_pair_u32>()))) return rc;
    SA_HIP_CHECK(hipMemcpyAsync(out, idx->q_out.p, (size_t)Q * sizeof(sa_hip_pair_u32), hipMemcpyDeviceToHost, idx->stream));
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    return 0;
}

// (idx->mu held)
static int query_hits_locked(sa_hip_index* idx, const uint8_t* pattern, uint64_t len, uint32_t max_hits,
                             sa_hip_pair_u32* range, uint32_t* hits, uint32_t* nhits) {
    if (!range || !nhits || (!pattern && len) || (!hits && max_hits)) return fail(SA_HIP_EINVAL, "sa_hip_index_query_hits: NULL argument");
    if (len > QH_BYTES - 64 - QH_OFF_PATTERN) return fail(SA_HIP_EINVAL, "sa_hip_index_query_hits: pattern longer than 47 KB");
    if (max_hits > QH_MAX_HITS) max_hits = QH_MAX_HITS;
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_query_hits: no index");
    if (idx->host) {
        const HostIndex& h = *idx->host;
        *range = h.query(pattern, len);
        u32 count = 0;
        if (range->first != 0xFFFFFFFFu && (u32)(range->second - range->first + 1u) != 0u) count = range->second - range->first + 1u;
        if (count > max_hits) count = max_hits;
        for (u32 i = 0; i < count; ++i) hits[i] = h.sa[(u64)range->first + i];
        *nhits = count;
        return 0;
    }
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (!idx->qh_host) {
        SA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&idx->qh_host), QH_BYTES, hipHostMallocMapped));
        SA_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&idx->qh_dev), idx->qh_host, 0));
    }
    u8* h = idx->qh_host;
    u8* d = idx->qh_dev;
    const u64 off[2] = {0, len};
    memcpy(h + QH_OFF_OFFSETS, off, sizeof off);
    if (len) memcpy(h + QH_OFF_PATTERN, pattern, len);
    memset(h + QH_OFF_PATTERN + len, 0, 64);
    if (idx->b.sector_search == 2) {   // the product configuration: one launch
        const QueryArgs qa = query_args(idx, d + QH_OFF_PATTERN, reinterpret_cast<const u64*>(d + QH_OFF_OFFSETS), 1, reinterpret_cast<sa_hip_pair_u32*>(d), 0);
        if (qa.keys32) hipLaunchKernelGGL(query_hits_one_kernel<true>, dim3(1), dim3(256), 0, idx->stream, qa, idx->b.qmap, (const u32*)idx->b.sa,
                                          max_hits, reinterpret_cast<u32*>(d + QH_OFF_HITS), reinterpret_cast<u32*>(d + 8));
        else hipLaunchKernelGGL(query_hits_one_kernel<false>, dim3(1), dim3(256), 0, idx->stream, qa, idx->b.qmap, (const u32*)idx->b.sa,
                                max_hits, reinterpret_cast<u32*>(d + QH_OFF_HITS), reinterpret_cast<u32*>(d + 8));
    } else {
        if ((rc = launch_query(idx, d + QH_OFF_PATTERN, reinterpret_cast<const u64*>(d + QH_OFF_OFFSETS), 1,
                               reinterpret_cast<sa_hip_pair_u32*>(d)))) return rc;
        hipLaunchKernelGGL(hits_copy_kernel, dim3(1), dim3(256), 0, idx->stream, (const u32*)idx->b.sa,
                           reinterpret_cast<const sa_hip_pair_u32*>(d), max_hits, reinterpret_cast<u32*>(d + QH_OFF_HITS),
                           reinterpret_cast<u32*>(d + 8));
    }
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    memcpy(range, h, sizeof *range);
    memcpy(nhits, h + 8, 4);
    if (*nhits) memcpy(hits, h + QH_OFF_HITS, (size_t)*nhits * 4);
    return 0;
}

int sa_hip_index_query_hits(sa_hip_index* idx, const uint8_t* pattern, uint64_t len, uint32_t max_hits,
                            sa_hip_pair_u32* range, uint32_t* hits, uint32_t* nhits) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_index_query_hits: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    return query_hits_locked(idx, pattern, len, max_hits, range, hits, nhits);
}

int sa_hip_query_batch_device(sa_hip_index* idx, const void* patterns_dev, const void* offsets_dev, uint64_t Q,
                              void* out_dev) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_query_batch_device: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_query_batch_device: the host path (no HIP device) has no device buffers");
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_query_batch_device: no index");
    if (Q == 0) return 0;
    if (!patterns_dev || !offsets_dev || !out_dev) return fail(SA_HIP_EINVAL, "sa_hip_query_batch_device: NULL argument");
    int rc = set_device(idx->device);
    if (rc) return rc;
    return launch_query(idx, (const u8*)patterns_dev, (const u64*)offsets_dev, Q, (sa_hip_pair_u32*)out_dev);
}

int sa_hip_query_batch_device_fixed(sa_hip_index* idx, const void* patterns_dev, uint64_t pattern_len, uint64_t Q, void* out_dev) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_query_batch_device_fixed: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) return fail(SA_HIP_EHIP, "sa_hip_query_batch_device_fixed: the host path (no HIP device) has no device buffers");
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_query_batch_device_fixed: no index");
    if (Q == 0) return 0;
    if ((!patterns_dev && pattern_len) || !out_dev) return fail(SA_HIP_EINVAL, "sa_hip_query_batch_device_fixed: NULL argument");
    int rc = set_device(idx->device);
    if (rc) return rc;
    return launch_query(idx, (const u8*)patterns_dev, nullptr, Q, (sa_hip_pair_u32*)out_dev, pattern_len);
}

int sa_hip_index_build_stats(const sa_hip_index* idx_c, sa_hip_build_stats* out) {
    sa_hip_index* idx = const_cast<sa_hip_index*>(idx_c);
    if (!idx || !out) return fail(SA_HIP_EINVAL, "sa_hip_index_build_stats: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) { memset(out, 0, sizeof *out); out->n = idx->host->n; return 0; }
    if (idx->widen_ms < 0.0) {
        int rc = set_device(idx->device);
        if (rc) return rc;
        SA_HIP_CHECK(hipEventSynchronize(idx->w_end));
        float ms = 0.f;
        SA_HIP_CHECK(hipEventElapsedTime(&ms, idx->w_begin, idx->w_end));
        idx->widen_ms = ms;
    }
    *out = idx->b.stats;
    out->widen_ms = idx->widen_ms;
    return 0;
}

int sa_hip_index_query_stats(const sa_hip_index* idx_c, sa_hip_query_stats* out) {
    sa_hip_index* idx = const_cast<sa_hip_index*>(idx_c);
    if (!idx || !out) return fail(SA_HIP_EINVAL, "sa_hip_index_query_stats: NULL argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host) { memset(out, 0, sizeof *out); return 0; }
    int rc = set_device(idx->device);
    if (rc) return rc;
    if ((rc = resolve_query_events(idx, sa_hip_index::QRING))) return rc;
    idx->qstats.kernel_ms = idx->q_last_ms;
    idx->qstats.kernel_ms_sum = idx->q_sum_ms;
    idx->qstats.launches = idx->q_launches;
    *out = idx->qstats;
    idx->q_sum_ms = 0.0;
    idx->q_launches = 0;
    return 0;
}

// ---- record retrieval (engine.c:920-999, 1168-1215, 1326-1390; pyx:87-101) -----------------------

static int set_rows_locked_tail(sa_hip_index* idx);

// the row table of the library's own CSV extractor: a malloc'ed array that is ascending by construction -- adopted, not copied
static int set_rows_adopt(sa_hip_index* idx, uint64_t* malloced_starts, uint64_t num_rows) {
    std::lock_guard<std::mutex> g(idx->mu);
    idx->row_starts.adopt(malloced_starts, (size_t)num_rows);
    return set_rows_locked_tail(idx);
}

int sa_hip_index_set_rows(sa_hip_index* idx, const uint64_t* row_text_starts, uint64_t num_rows) {
    if (!idx || (!row_text_starts && num_rows)) return fail(SA_HIP_EINVAL, "sa_hip_index_set_rows: NULL argument");
    if (num_rows && row_text_starts[0] != 0) return fail(SA_HIP_EINVAL, "sa_hip_index_set_rows: the first row must start at offset 0");
    for (uint64_t r = 1; r < num_rows; ++r)
        if (row_text_starts[r] < row_text_starts[r - 1]) return fail(SA_HIP_EINVAL, "sa_hip_index_set_rows: offsets must ascend");
    std::lock_guard<std::mutex> g(idx->mu);
    try { idx->row_starts.assign(row_text_starts, row_text_starts + num_rows); }
    catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_index_set_rows: out of host memory"); }
    return set_rows_locked_tail(idx);
}

static int set_rows_locked_tail(sa_hip_index* idx) {   // (idx->mu held) the table in HBM for the rows kernel
    const u64* row_text_starts = idx->row_starts.data();
    const u64 num_rows = idx->row_starts.size();
    if (idx->host) return 0;
    // ... and in HBM for the rows kernel
    int rc = set_device(idx->device);
    if (rc) return rc;
    if ((rc = idx->rows_dev.ensure((size_t)(num_rows ? num_rows : 1) * 8))) return rc;
    if (num_rows) SA_HIP_CHECK(hipMemcpyAsync(idx->rows_dev.p, row_text_starts, (size_t)num_rows * 8, hipMemcpyHostToDevice, idx->stream));
    // every 256th start as a table of its own: the rows kernel's search runs through it first
    std::vector<u64> coarse;
    try {
        for (u64 r = 0; r < num_rows; r += (1ull << ROWS_COARSE_SHIFT)) coarse.push_back(row_text_starts[r]);
    } catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_index_set_rows: out of host memory"); }
    idx->rows_coarse_n = coarse.size();
    if ((rc = idx->rows_coarse.ensure((coarse.size() ? coarse.size() : 1) * 8))) return rc;
    if (!coarse.empty()) SA_HIP_CHECK(hipMemcpyAsync(idx->rows_coarse.p, coarse.data(), coarse.size() * 8, hipMemcpyHostToDevice, idx->stream));
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));   // (the staging vector lives until here)
    return 0;
}

int sa_hip_index_get_text(sa_hip_index* idx, uint8_t* out_host) {
    if (!idx) return fail(SA_HIP_EINVAL, "sa_hip_index_get_text: NULL index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_get_text: no index");
    if (!out_host && sa_hip_index_n(idx)) return fail(SA_HIP_EINVAL, "sa_hip_index_get_text: NULL output");
    if (idx->host) { if (idx->host->n) memcpy(out_host, idx->host->text.data(), (size_t)idx->host->n); return 0; }
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (idx->b.n) SA_HIP_CHECK(hipMemcpyAsync(out_host, idx->b.text.p, (size_t)idx->b.n, hipMemcpyDeviceToHost, idx->stream));
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    return 0;
}

// host path of the record retrieval (records.hpp): k beyond ROWS_K_MAX ("all rows"), and SA_HIP_HOST_ROWS=1 (diagnostic:
// the tests compare the two paths row for row)
static bool host_rows_forced() {
    const char* e = diag_env("SA_HIP_HOST_ROWS");
    return e && atoi(e) != 0;
}

// (idx->mu held) ONE query: search + rows kernel through the pinned block, one synchronisation
static int query_rows_device_locked(sa_hip_index* idx, const uint8_t* pattern, uint64_t len, uint32_t k, uint64_t* row_ids,
                                    uint32_t* num_rows, sa_hip_pair_u32* range) {
    if (len > QH_BYTES - 64 - QH_OFF_PATTERN) return fail(SA_HIP_EINVAL, "sa_hip_index_query_rows: pattern longer than 47 KB");
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_query_rows: no index");
    int rc = set_device(idx->device);
    if (rc) return rc;
    if (!idx->qh_host) {
        SA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&idx->qh_host), QH_BYTES, hipHostMallocMapped));
        SA_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&idx->qh_dev), idx->qh_host, 0));
    }
    u8* h = idx->qh_host;
    u8* d = idx->qh_dev;
    const u64 off[2] = {0, len};
    memcpy(h + QH_OFF_OFFSETS, off, sizeof off);
    if (len) memcpy(h + QH_OFF_PATTERN, pattern, len);
    memset(h + QH_OFF_PATTERN + len, 0, 64);
    RowsArgs a;
    a.sa = idx->b.sa; a.ranges = reinterpret_cast<const sa_hip_pair_u32*>(d); a.q = 1;
    a.row_starts = idx->rows_dev.as<u64>(); a.num_rows = idx->row_starts.size(); a.k = k;
    a.coarse = (idx->rows_coarse_n > 1) ? idx->rows_coarse.as<u64>() : nullptr; a.coarse_n = idx->rows_coarse_n;
    a.out_rows = reinterpret_cast<u32*>(d + QH_OFF_HITS); a.out_counts = reinterpret_cast<u32*>(d + 8);
    if (idx->b.sector_search == 2 && k <= ROWS_K_SMALL) {
        // the product configuration: search + rows in ONE launch (no events: this path is not part of the query statistics)
        const QueryArgs qa = query_args(idx, d + QH_OFF_PATTERN, reinterpret_cast<const u64*>(d + QH_OFF_OFFSETS), 1, reinterpret_cast<sa_hip_pair_u32*>(d), 0);
        if (qa.keys32) hipLaunchKernelGGL((query_rows_one_kernel<true, ROWS_SLOTS_SMALL>), dim3(1), dim3(256), 0, idx->stream, qa, idx->b.qmap, a);
        else hipLaunchKernelGGL((query_rows_one_kernel<false, ROWS_SLOTS_SMALL>), dim3(1), dim3(256), 0, idx->stream, qa, idx->b.qmap, a);
    } else {
        if ((rc = launch_query(idx, d + QH_OFF_PATTERN, reinterpret_cast<const u64*>(d + QH_OFF_OFFSETS), 1,
                               reinterpret_cast<sa_hip_pair_u32*>(d)))) return rc;
        launch_rows(idx->stream, a);
    }
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    if (range) memcpy(range, h, sizeof *range);
    u32 n = 0;
    memcpy(&n, h + 8, 4);
    const u32* rows = reinterpret_cast<const u32*>(h + QH_OFF_HITS);
    for (u32 i = 0; i < n; ++i) row_ids[i] = rows[i];
    *num_rows = n;
    return 0;
}

int sa_hip_index_query_rows(sa_hip_index* idx, const uint8_t* pattern, uint64_t len, uint32_t k, uint64_t* row_ids,
                            uint32_t* num_rows, sa_hip_pair_u32* range) {
    if (!idx || !num_rows || (!row_ids && k) || (!pattern && len)) return fail(SA_HIP_EINVAL, "sa_hip_index_query_rows: NULL argument");
    *num_rows = 0;
    sa_hip_pair_u32 rg;
    u32 nh = 0;
    std::lock_guard<std::mutex> g(idx->mu);   // the search, the slabs of hits and the row table under ONE acquisition
    if (k && idx->row_starts.empty()) return fail(SA_HIP_EINVAL, "sa_hip_index_query_rows: no row table (sa_hip_index_set_rows)");
    // more rows than the table holds cannot come back: k = 10^9 ("all") must not size anything
    if ((u64)k > idx->row_starts.size()) k = (u32)idx->row_starts.size();
    if (k && k <= ROWS_K_MAX && !host_rows_forced() && !idx->host) return query_rows_device_locked(idx, pattern, len, k, row_ids, num_rows, range);
    const u32 cap = k ? std::min<u32>(std::max<u32>(4u * k, 1024u), QH_MAX_HITS) : 0u;
    try {
        std::vector<u32> first(cap ? cap : 1);
        int rc = query_hits_locked(idx, pattern, len, cap, &rg, first.data(), &nh);
        if (rc) return rc;
        if (range) *range = rg;
        const HostU64Array& starts = idx->row_starts;
        if (starts.empty()) return 0;
        std::vector<u64> rows;
        rc = distinct_rows(starts, rg, k, first.data(), nh,
                           [&](u64 pos, u64 count, u32* out) { return get_sa_range_locked(idx, pos, count, out); }, rows);
        if (rc) return rc;
        for (size_t i = 0; i < rows.size(); ++i) row_ids[i] = rows[i];
        *num_rows = (uint32_t)rows.size();
    } catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_index_query_rows: out of host memory"); }
    return 0;
}

// (batches whose Q x k row ids take less than this go down by one plain copy; SA_HIP_ROWS_RING=0: always -- A/B, tests)
static constexpr size_t ROWS_RING_MIN_BYTES = 32u << 20;
static bool rows_waves_off() { const char* e = diag_env("SA_HIP_ROWS_WAVES"); return e && e[0] == '0'; }   // A/B, tests: no wave-per-range form
static u32 rows_wave_groups() { const char* e = diag_env("SA_HIP_ROWS_WAVE_GROUPS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 256 ? (u32)v : 8u; }   // A/B: workgroups of the wave form per CU
static bool rows_lanes_off() { const char* e = diag_env("SA_HIP_ROWS_LANES"); return e && e[0] == '0'; }   // A/B, tests: every query through the workgroup form
static bool rows_trace_on() { const char* e = diag_env("SA_HIP_ROWS_TRACE"); return e && e[0] == '1'; }   // one stderr line per large batch: where its time went
static bool rows_ring_off() { const char* e = diag_env("SA_HIP_ROWS_RING"); return e && e[0] == '0'; }

int sa_hip_index_query_rows_batch(sa_hip_index* idx, const uint8_t* patterns, const uint64_t* offsets, uint64_t Q, uint32_t k,
                                  uint64_t* row_ids, uint32_t* counts, sa_hip_pair_u32* ranges) {
    if (!idx || (Q && (!offsets || !counts || (!row_ids && k)))) return fail(SA_HIP_EINVAL, "sa_hip_index_query_rows_batch: NULL argument");
    if (Q == 0) return 0;
    std::lock_guard<std::mutex> g(idx->mu);
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_query_rows_batch: no index");
    const u64 total = offsets[Q];
    if (!patterns && total) return fail(SA_HIP_EINVAL, "sa_hip_index_query_rows_batch: NULL patterns");
    const u32 k_in = k;
    if (k && idx->row_starts.empty()) return fail(SA_HIP_EINVAL, "sa_hip_index_query_rows_batch: no row table (sa_hip_index_set_rows)");
    if ((u64)k > idx->row_starts.size()) k = (u32)idx->row_starts.size();
    for (u64 i = 0; i < Q; ++i) counts[i] = 0;
    if (idx->host) {
        try {
            std::vector<u64> rows;
            for (u64 q = 0; q < Q; ++q) {
                const sa_hip_pair_u32 rg = idx->host->query(patterns + offsets[q], offsets[q + 1] - offsets[q]);
                if (ranges) ranges[q] = rg;
                if (!k) continue;
                int rc = distinct_rows(idx->row_starts, rg, k, nullptr, 0,
                                       [&](u64 pos, u64 count, u32* out) { return get_sa_range_locked(idx, pos, count, out); }, rows);
                if (rc) return rc;
                for (size_t i = 0; i < rows.size(); ++i) row_ids[q * k_in + i] = rows[i];
                counts[q] = (u32)rows.size();
            }
        } catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_index_query_rows_batch: out of host memory"); }
        return 0;
    }
    int rc = set_device(idx->device);
    if (rc) return rc;
    // ONE search launch for all the ranges
    if ((rc = idx->q_pat.ensure((size_t)total + 64))) return rc;
    if ((rc = idx->q_off.ensure((size_t)(Q + 1) * 8))) return rc;
    if ((rc = idx->q_out.ensure((size_t)Q * sizeof(sa_hip_pair_u32)))) return rc;
    const bool device_rows = k && k <= ROWS_K_MAX && !host_rows_forced();
    // A large batch moves its host legs through the process's ring of pinned slabs (host_io.hpp): patterns and offsets up, counts,
    // ranges and the Q x k row ids down as u32, widened into the caller's uint64[Q][k] by the ring's worker threads while the
    // next slabs arrive -- instead of one pageable copy into a Q x k vector (allocated and page-faulted per call) and one
    // thread widening it (1e7 patterns, k = 16: 353 -> see profiles/r04_n_rows_batch.log).  Same bytes, same results.
    std::unique_lock<std::mutex> ring_lock;
    PinnedRing* ring = nullptr;
    if (device_rows && (size_t)Q * k * 4 >= ROWS_RING_MIN_BYTES && !rows_ring_off()) {
        ring_lock = std::unique_lock<std::mutex>(g_oneshot.mu);
        if (!g_oneshot.ring.ready && (rc = g_oneshot.ring.init())) return rc;
        if (g_oneshot.ring.device == idx->device) ring = &g_oneshot.ring;
        else ring_lock.unlock();   // (the ring's copy stream lives on another device: the plain copies below)
    }
    const bool trace = ring && rows_trace_on();
    auto t_mark = std::chrono::steady_clock::now();
    double t_up = 0, t_kern = 0, t_small = 0, t_rows = 0;
    auto lap = [&](double& acc) { const auto now = std::chrono::steady_clock::now(); acc += std::chrono::duration<double, std::milli>(now - t_mark).count(); t_mark = now; };
    if (ring) {
        if ((rc = ring_upload(*ring, idx->stream, idx->device, idx->q_pat.p, patterns, (size_t)total))) return rc;
        if ((rc = ring_upload(*ring, idx->stream, idx->device, idx->q_off.p, reinterpret_cast<const u8*>(offsets), (size_t)(Q + 1) * 8))) return rc;
        lap(t_up);
    } else {
        if (total) SA_HIP_CHECK(hipMemcpyAsync(idx->q_pat.p, patterns, total, hipMemcpyHostToDevice, idx->stream));
        SA_HIP_CHECK(hipMemcpyAsync(idx->q_off.p, offsets, (size_t)(Q + 1) * 8, hipMemcpyHostToDevice, idx->stream));
    }
    SA_HIP_CHECK(hipMemsetAsync(idx->q_pat.as<u8>() + total, 0, 64, idx->stream));
    if ((rc = launch_query(idx, idx->q_pat.as<u8>(), idx->q_off.as<u64>(), Q, idx->q_out.as<sa_hip_pair_u32>()))) return rc;
    try {
        std::vector<sa_hip_pair_u32> rg_host;
        if (!ring && (ranges || !device_rows)) {
            rg_host.resize(Q);
            SA_HIP_CHECK(hipMemcpyAsync(rg_host.data(), idx->q_out.p, (size_t)Q * sizeof(sa_hip_pair_u32), hipMemcpyDeviceToHost, idx->stream));
        }
        if (device_rows) {
            // ... ONE rows launch for all the hits -> rows, one copy back: no per-query synchronisation
            if ((rc = idx->r_rows.ensure((size_t)Q * k * 4))) return rc;
            if ((rc = idx->r_counts.ensure((size_t)Q * 4))) return rc;
            const bool lanes = Q >= ROWS_LANE_MIN_BATCH && !rows_lanes_off();
            if (lanes && (rc = idx->r_pending.ensure((2 * (size_t)Q + 2) * 4))) return rc;
            RowsArgs a;
            a.sa = idx->b.sa; a.ranges = idx->q_out.as<sa_hip_pair_u32>(); a.q = Q;
            a.row_starts = idx->rows_dev.as<u64>(); a.num_rows = idx->row_starts.size(); a.k = k;
            a.coarse = (idx->rows_coarse_n > 1) ? idx->rows_coarse.as<u64>() : nullptr; a.coarse_n = idx->rows_coarse_n;
            a.out_rows = idx->r_rows.as<u32>(); a.out_counts = idx->r_counts.as<u32>();
            launch_rows(idx->stream, a, lanes ? idx->r_pending.as<u32>() : nullptr, !rows_waves_off(), rows_wave_groups());
            SA_HIP_CHECK(hipGetLastError());
            if (ring) {
                SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
                lap(t_kern);
                if ((rc = ring_download<u32>(*ring, idx->device, idx->r_counts.as<u32>(), counts, (size_t)Q))) return rc;
                if (ranges && (rc = ring_download<u32>(*ring, idx->device, idx->q_out.as<u32>(), reinterpret_cast<u32*>(ranges), (size_t)Q * 2))) return rc;
                lap(t_small);
                const size_t per_q = (size_t)k * 4;                                        // k <= ROWS_K_MAX: at most 16 KB
                // whole queries per piece; about 32 pieces (1 MB .. one slab) so that a mid-size batch still spreads over all workers
                const size_t target = std::min<size_t>(PinnedRing::SLAB_BYTES, std::max<size_t>((size_t)1 << 20, (size_t)Q * per_q / 32));
                const size_t piece = std::max<size_t>(per_q, (target / per_q) * per_q);
                rc = ring_download_pieces(*ring, idx->device, idx->r_rows.as<u8>(), (size_t)Q * per_q, piece,
                                          [&](const u8* p, size_t off, size_t len) {
                                              const u32* in = reinterpret_cast<const u32*>(p);
                                              const u64 q0 = off / per_q, nq = len / per_q;
                                              for (u64 j = 0; j < nq; ++j) {
                                                  const u32 c = counts[q0 + j];
                                                  u64* o = row_ids + (q0 + j) * k_in;
                                                  const u32* r = in + j * k;
                                                  for (u32 i = 0; i < c; ++i) o[i] = r[i];
                                              }
                                          });
                lap(t_rows);
                if (trace) fprintf(stderr, "[sa_hip rows batch] Q=%llu k=%u: patterns+offsets up %.2f ms, search+rows kernels %.2f, counts+ranges down %.2f, row ids down+widened %.2f (pieces of %zu bytes)\n",
                                   (unsigned long long)Q, k, t_up, t_kern, t_small, t_rows, piece);
                return rc;
            }
            std::vector<u32> rows32((size_t)Q * k);
            SA_HIP_CHECK(hipMemcpyAsync(counts, idx->r_counts.p, (size_t)Q * 4, hipMemcpyDeviceToHost, idx->stream));
            SA_HIP_CHECK(hipMemcpyAsync(rows32.data(), idx->r_rows.p, (size_t)Q * k * 4, hipMemcpyDeviceToHost, idx->stream));
            SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
            for (u64 q = 0; q < Q; ++q)
                for (u32 i = 0; i < counts[q]; ++i) row_ids[q * k_in + i] = rows32[q * k + i];
        } else {
            SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
            if (k) {
                std::vector<u64> rows;
                for (u64 q = 0; q < Q; ++q) {
                    rc = distinct_rows(idx->row_starts, rg_host[q], k, nullptr, 0,
                                       [&](u64 pos, u64 count, u32* out) { return get_sa_range_locked(idx, pos, count, out); }, rows);
                    if (rc) return rc;
                    for (size_t i = 0; i < rows.size(); ++i) row_ids[q * k_in + i] = rows[i];
                    counts[q] = (u32)rows.size();
                }
            }
        }
        if (ranges) memcpy(ranges, rg_host.data(), (size_t)Q * sizeof(sa_hip_pair_u32));
    } catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_index_query_rows_batch: out of host memory"); }
    return 0;
}

int sa_hip_index_rows_for_range(sa_hip_index* idx, sa_hip_pair_u32 range, uint32_t k, uint64_t* row_ids, uint32_t* num_rows) {
    if (!idx || !num_rows || (!row_ids && k)) return fail(SA_HIP_EINVAL, "sa_hip_index_rows_for_range: NULL argument");
    *num_rows = 0;
    std::lock_guard<std::mutex> g(idx->mu);
    if (!idx->has_index) return fail(SA_HIP_EINVAL, "sa_hip_index_rows_for_range: no index");
    if (range.first != 0xFFFFFFFFu && (u32)(range.second - range.first + 1u) != 0u &&
        ((u64)range.second >= (idx->host ? idx->host->n : idx->b.n) || range.first > range.second)) return fail(SA_HIP_EINVAL, "sa_hip_index_rows_for_range: range outside the suffix array");
    if (k && idx->row_starts.empty()) return fail(SA_HIP_EINVAL, "sa_hip_index_rows_for_range: no row table (sa_hip_index_set_rows)");
    if ((u64)k > idx->row_starts.size()) k = (u32)idx->row_starts.size();
    try {
        const HostU64Array& starts = idx->row_starts;
        if (starts.empty()) return 0;
        std::vector<u64> rows;
        int rc = distinct_rows(starts, range, k, nullptr, 0,
                               [&](u64 pos, u64 count, u32* out) { return get_sa_range_locked(idx, pos, count, out); }, rows);
        if (rc) return rc;
        for (size_t i = 0; i < rows.size(); ++i) row_ids[i] = rows[i];
        *num_rows = (uint32_t)rows.size();
    } catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_index_rows_for_range: out of host memory"); }
    return 0;
}

static int csv_index_finish(sa_hip_csv_index* c) {
    const int fd = open(c->path.c_str(), O_RDONLY);
    if (fd < 0) return fail(SA_HIP_EINVAL, "sa_hip_csv_index: cannot open file", c->path.c_str());
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return fail(SA_HIP_EINVAL, "sa_hip_csv_index: fstat failed", c->path.c_str()); }
    c->map_len = (u64)sb.st_size;
    if (!c->row_file_offsets.empty() && c->row_file_offsets.back() > c->map_len) { close(fd); return fail(SA_HIP_EINVAL, "sa_hip_csv_index: row table exceeds the file"); }
    void* m = c->map_len ? mmap(nullptr, c->map_len, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (c->map_len && m == MAP_FAILED) return fail(SA_HIP_ENOMEM, "sa_hip_csv_index: mmap failed", c->path.c_str());
    c->map = static_cast<const u8*>(m);
    return 0;
}

void sa_hip_csv_index_destroy(sa_hip_csv_index* c) {
    if (!c) return;
    if (c->map) munmap(const_cast<u8*>(c->map), c->map_len);
    if (c->idx) sa_hip_index_destroy(c->idx);
    delete c;
}

int sa_hip_csv_index_create(sa_hip_csv_index** out, const char* csv_file, const char* search_column, uint32_t max_suffix_length,
                            int device) {
    if (!out || !csv_file || !search_column) return fail(SA_HIP_EINVAL, "sa_hip_csv_index_create: NULL argument");
    *out = nullptr;
    if (max_suffix_length == 0) return fail(SA_HIP_EINVAL, "sa_hip_csv_index_create: max_suffix_length must be >= 1");
    sa_hip_csv_column col;
    // the HIP runtime of this process (context, code object: ~0.2 s the first time) comes up while the host parses
    std::thread warm;
    try { warm = std::thread([device]() { if (hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr); }); } catch (...) {}
    int rc = sa_hip_csv_extract_column(csv_file, search_column, &col);
    if (warm.joinable()) warm.join();
    if (rc) return rc;
    sa_hip_csv_index* c = nullptr;
    try {
        c = new sa_hip_csv_index();
        c->path = csv_file;
        c->column_index = col.column_index;
        const char* p = col.column_names;
        for (u32 i = 0; i < col.num_columns; ++i) { c->columns.emplace_back(p); p += c->columns.back().size() + 1; }
        c->row_file_offsets.assign(col.row_file_offsets, col.row_file_offsets + col.num_rows + 1);
    } catch (const std::bad_alloc&) {
        delete c; sa_hip_csv_free(&col);
        return fail(SA_HIP_ENOMEM, "sa_hip_csv_index_create: out of host memory");
    }
    if (col.text_len > 0xFFFFFFFEull) rc = fail(SA_HIP_EINVAL, "sa_hip_csv_index_create: the column exceeds 2^32 - 2 bytes (one index per device)");
    if (!rc) rc = sa_hip_index_create(&c->idx, col.text_len ? col.text_len : 1, device);
    if (!rc) rc = sa_hip_index_build(c->idx, col.text, col.text_len, max_suffix_length);
    if (!rc) rc = sa_hip_index_set_rows(c->idx, col.row_text_starts, col.num_rows);
    sa_hip_csv_free(&col);
    if (!rc) rc = csv_index_finish(c);
    if (rc) { sa_hip_csv_index_destroy(c); return rc; }
    *out = c;
    return 0;
}

// one helper thread at a time gives large host buffers back (joined before the next one starts and when the library is unloaded:
// never detached)
namespace {
struct BackgroundFree {
    std::thread t;
    std::mutex m;
    void run(sa_hip_csv_column col) {
        std::lock_guard<std::mutex> g(m);
        if (t.joinable()) t.join();
        try { t = std::thread([col]() mutable { csv_free(&col); }); } catch (...) { csv_free(&col); }
    }
    ~BackgroundFree() { if (t.joinable()) t.join(); }
};
BackgroundFree g_bgfree;
}  // namespace

// The reference cuts a CSV file into 2 GiB partitions, one suffix array each (engine.c:1437-1481), and answers a query from
// the partitions one after the other (suffix_array.pyx:221-247).  Here one index holds up to 2^32 - 2 COLUMN bytes whatever
// the file's size; a column beyond `partition_bytes` is cut at ROW boundaries (no match is lost at a cut) into several
// independent sa_hip_csv_index objects over the same mapped file -- every entry point works on a part as on a whole index.
int sa_hip_csv_index_create_partitioned(sa_hip_csv_index*** out_parts, uint32_t* num_parts, const char* csv_file, const char* search_column,
                                        uint32_t max_suffix_length, int device, uint64_t partition_bytes) {
    if (!out_parts || !num_parts || !csv_file || !search_column) return fail(SA_HIP_EINVAL, "sa_hip_csv_index_create_partitioned: NULL argument");
    *out_parts = nullptr; *num_parts = 0;
    if (max_suffix_length == 0) return fail(SA_HIP_EINVAL, "sa_hip_csv_index_create_partitioned: max_suffix_length must be >= 1");
    if (partition_bytes == 0 || partition_bytes > 0xFFFFFFFEull) partition_bytes = 0xFFFFFFFEull;
    sa_hip_csv_column col;
    std::thread warm;
    try { warm = std::thread([device]() { if (hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr); }); } catch (...) {}
    const bool timing = diag_env("SA_HIP_CSV_TIMING") && atoi(diag_env("SA_HIP_CSV_TIMING")) != 0;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sa_hip csv] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    int rc = sa_hip_csv_extract_column(csv_file, search_column, &col);
    if (warm.joinable()) warm.join();
    if (rc) return rc;
    lap("extract (all of the above)");
    // rows [cut[p], cut[p + 1]) form part p: as many whole rows as fit partition_bytes (a row's text includes its separator)
    std::vector<u64> cut;
    std::vector<sa_hip_csv_index*> parts;
    try {
        cut.push_back(0);
        u64 begin_text = 0;
        // (the usual case -- the whole column fits one index -- needs no walk over the rows)
        for (u64 r = 0; col.text_len > partition_bytes && r < col.num_rows; ++r) {
            const u64 row_end = (r + 1 < col.num_rows) ? col.row_text_starts[r + 1] : col.text_len;
            if (row_end - col.row_text_starts[r] > partition_bytes) {
                sa_hip_csv_free(&col);
                return fail(SA_HIP_EINVAL, "sa_hip_csv_index_create_partitioned: one row's column text exceeds partition_bytes");
            }
            if (row_end - begin_text > partition_bytes) { cut.push_back(r); begin_text = col.row_text_starts[r]; }
        }
        cut.push_back(col.num_rows);
        if (cut.size() == 2 && col.num_rows == 0) cut.assign({0, 0});
        for (size_t p = 0; p + 1 < cut.size() && !rc; ++p) {
            const u64 r0 = cut[p], r1 = cut[p + 1];
            const u64 t0 = (r0 < col.num_rows) ? col.row_text_starts[r0] : col.text_len;
            const u64 t1 = (r1 < col.num_rows) ? col.row_text_starts[r1] : col.text_len;
            sa_hip_csv_index* c = new sa_hip_csv_index();
            parts.push_back(c);
            c->path = csv_file;
            c->column_index = col.column_index;
            const char* q = col.column_names;
            for (u32 i = 0; i < col.num_columns; ++i) { c->columns.emplace_back(q); q += c->columns.back().size() + 1; }
            const bool whole = (r0 == 0 && r1 == col.num_rows);   // one part: the extractor's tables are adopted as they are
            std::vector<u64> starts;
            if (whole) {
                c->row_file_offsets.adopt(col.row_file_offsets, (size_t)col.num_rows + 1);
                col.row_file_offsets = nullptr;
            } else {
                c->row_file_offsets.assign(col.row_file_offsets + r0, col.row_file_offsets + r1 + 1);
                starts.resize(r1 - r0);
                for (u64 r = r0; r < r1; ++r) starts[r - r0] = col.row_text_starts[r] - t0;
            }
            const u64 len = t1 - t0;
            lap("cuts, row tables");
            rc = sa_hip_index_create(&c->idx, len ? len : 1, device);
            lap("index create (buffers)");
            if (!rc) rc = sa_hip_index_build(c->idx, col.text + t0, len, max_suffix_length);
            lap("upload + build");
            if (!rc) {
                if (whole) { u64* own = col.row_text_starts; col.row_text_starts = nullptr; rc = set_rows_adopt(c->idx, own, col.num_rows); }
                else rc = sa_hip_index_set_rows(c->idx, starts.data(), r1 - r0);
            }
            lap("row table to the device");
            if (!rc) rc = csv_index_finish(c);
            lap("file mapping");
        }
    } catch (const std::bad_alloc&) {
        rc = fail(SA_HIP_ENOMEM, "sa_hip_csv_index_create_partitioned: out of host memory");
    }
    // the extracted column (0.9 GB of touched pages at 50M rows) is given back on a thread of its own: unmapping it takes 0.2 s
    g_bgfree.run(col);
    memset(&col, 0, sizeof col);
    lap("free the extracted column");
    sa_hip_csv_index** arr = rc ? nullptr : static_cast<sa_hip_csv_index**>(malloc((parts.size() ? parts.size() : 1) * sizeof(sa_hip_csv_index*)));
    if (!rc && !arr) rc = fail(SA_HIP_ENOMEM, "sa_hip_csv_index_create_partitioned: out of host memory");
    if (rc) { for (sa_hip_csv_index* c : parts) sa_hip_csv_index_destroy(c); return rc; }
    for (size_t p = 0; p < parts.size(); ++p) arr[p] = parts[p];
    *out_parts = arr;
    *num_parts = (uint32_t)parts.size();
    return 0;
}
void sa_hip_csv_index_free_parts(sa_hip_csv_index** parts) { free(parts); }

int sa_hip_csv_index_adopt(sa_hip_csv_index** out, const char* csv_file, const uint8_t* text, const uint32_t* SA, uint64_t n,
                           const uint64_t* row_text_starts, const uint64_t* row_file_offsets, uint64_t num_rows,
                           const char* column_names, uint32_t num_columns, uint32_t column_index, uint32_t max_suffix_length, int device) {
    if (!out || !csv_file || (!text && n) || (!SA && n) || (num_rows && (!row_text_starts || !row_file_offsets)) || (!column_names && num_columns))
        return fail(SA_HIP_EINVAL, "sa_hip_csv_index_adopt: NULL argument");
    *out = nullptr;
    sa_hip_csv_index* c = nullptr;
    try {
        c = new sa_hip_csv_index();
        c->path = csv_file;
        c->column_index = column_index;
        const char* p = column_names;
        for (u32 i = 0; i < num_columns; ++i) { c->columns.emplace_back(p); p += c->columns.back().size() + 1; }
        if (num_rows) c->row_file_offsets.assign(row_file_offsets, row_file_offsets + num_rows + 1);
    } catch (const std::bad_alloc&) { delete c; return fail(SA_HIP_ENOMEM, "sa_hip_csv_index_adopt: out of host memory"); }
    // the tables come from a file: a row is copied out of the mapping by [row_file_offsets[r], row_file_offsets[r + 1]) and found
    // by row_text_starts -- both must ascend and stay inside what they index (csv_index_finish checks the last offset
    // against the file's size)
    for (u64 r = 0; r < num_rows; ++r) {
        if (row_file_offsets[r] > row_file_offsets[r + 1]) { delete c; return fail(SA_HIP_EINVAL, "sa_hip_csv_index_adopt: row_file_offsets must ascend"); }
        if (row_text_starts[r] > n) { delete c; return fail(SA_HIP_EINVAL, "sa_hip_csv_index_adopt: row_text_starts beyond the text"); }
    }
    if (column_index >= num_columns && num_columns) { delete c; return fail(SA_HIP_EINVAL, "sa_hip_csv_index_adopt: column_index out of range"); }
    int rc = sa_hip_index_create(&c->idx, n ? n : 1, device);
    if (!rc) rc = sa_hip_index_load(c->idx, text, SA, n, max_suffix_length);
    if (!rc) rc = sa_hip_index_set_rows(c->idx, row_text_starts, num_rows);
    if (!rc) rc = csv_index_finish(c);
    if (rc) { sa_hip_csv_index_destroy(c); return rc; }
    *out = c;
    return 0;
}

sa_hip_index* sa_hip_csv_index_handle(sa_hip_csv_index* c) { return c ? c->idx : nullptr; }
uint64_t sa_hip_csv_index_num_rows(const sa_hip_csv_index* c) { return (c && !c->row_file_offsets.empty()) ? c->row_file_offsets.size() - 1 : 0; }
uint32_t sa_hip_csv_index_num_columns(const sa_hip_csv_index* c) { return c ? (uint32_t)c->columns.size() : 0; }
uint32_t sa_hip_csv_index_column_index(const sa_hip_csv_index* c) { return c ? c->column_index : 0; }
const char* sa_hip_csv_index_column_name(const sa_hip_csv_index* c, uint32_t i) { return (c && i < c->columns.size()) ? c->columns[i].c_str() : nullptr; }
int sa_hip_csv_index_row_tables(const sa_hip_csv_index* c, const uint64_t** row_text_starts, const uint64_t** row_file_offsets) {
    if (!c || !c->idx) return fail(SA_HIP_EINVAL, "sa_hip_csv_index_row_tables: NULL index");
    if (row_text_starts) *row_text_starts = c->idx->row_starts.data();
    if (row_file_offsets) *row_file_offsets = c->row_file_offsets.data();
    return 0;
}

int sa_hip_csv_index_copy_rows(sa_hip_csv_index* c, const uint64_t* row_ids, uint32_t n, char** records) {
    if (!c || (n && (!row_ids || !records))) return fail(SA_HIP_EINVAL, "sa_hip_csv_index_copy_rows: NULL argument");
    const u64 rows = sa_hip_csv_index_num_rows(c);
    for (u32 i = 0; i < n; ++i) records[i] = nullptr;
    for (u32 i = 0; i < n; ++i) {
        if (row_ids[i] >= rows) { sa_hip_free_records(records, i); return fail(SA_HIP_EINVAL, "sa_hip_csv_index_copy_rows: no such row"); }
        records[i] = dup_row(c->map, c->row_file_offsets[row_ids[i]], c->row_file_offsets[row_ids[i] + 1]);
        if (!records[i]) { sa_hip_free_records(records, i); return fail(SA_HIP_ENOMEM, "sa_hip_csv_index_copy_rows: out of host memory"); }
    }
    return 0;
}

sa_hip_pair_u32 sa_hip_get_substring_positions_file(sa_hip_csv_index* c, const char* substring) {
    sa_hip_pair_u32 r = {0xFFFFFFFFu, 0xFFFFFFFFu};
    if (!c || !c->idx || !substring) { fail(SA_HIP_EINVAL, "sa_hip_get_substring_positions_file: NULL argument"); return r; }
    u32 nh = 0;
    sa_hip_pair_u32 q;
    if (sa_hip_index_query_hits(c->idx, reinterpret_cast<const uint8_t*>(substring), strlen(substring), 0, &q, nullptr, &nh)) return r;
    // engine.c:962-965: start / end are only ever set on an exact match, so ANY miss is {UINT32_MAX, UINT32_MAX}
    if (q.first == 0xFFFFFFFFu || (u32)(q.second - q.first + 1u) == 0u) return r;
    return q;
}

int sa_hip_get_matching_records_file(sa_hip_csv_index* c, const char* substring, uint32_t k, char** matching_records,
                                     uint32_t* num_matches) {
    if (!c || !c->idx || !substring || !num_matches || (!matching_records && k)) return fail(SA_HIP_EINVAL, "sa_hip_get_matching_records_file: NULL argument");
    if (*num_matches >= k) return 0;   // engine.c:1356: at most k - *num_matches more
    // never more than the file has rows: k = 10^9 ("all") must not size a 8 GB table
    const u32 want = (u32)std::min<u64>(k - *num_matches, sa_hip_csv_index_num_rows(c));
    if (want == 0) return 0;
    try {
        std::vector<u64> rows((size_t)want);
        u32 n = 0;
        int rc = sa_hip_index_query_rows(c->idx, reinterpret_cast<const uint8_t*>(substring), strlen(substring), want, rows.data(), &n, nullptr);
        if (rc) return rc;
        for (u32 i = 0; i < n; ++i) {
            const u64 r = rows[i];
            char* s = dup_row(c->map, c->row_file_offsets[r], c->row_file_offsets[r + 1]);
            if (!s) return fail(SA_HIP_ENOMEM, "sa_hip_get_matching_records_file: out of host memory");
            matching_records[(*num_matches)++] = s;
        }
    } catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_get_matching_records_file: out of host memory"); }
    return 0;
}

int sa_hip_get_matching_row_spans_file(sa_hip_csv_index* c, const char* substring, uint32_t k, const char** row_ptrs,
                                       uint32_t* row_lens, uint32_t* num_matches) {
    if (!c || !c->idx || !substring || !num_matches || (k && (!row_ptrs || !row_lens))) return fail(SA_HIP_EINVAL, "sa_hip_get_matching_row_spans_file: NULL argument");
    *num_matches = 0;
    const u32 want = (u32)std::min<u64>(k, sa_hip_csv_index_num_rows(c));
    if (want == 0) return 0;
    try {
        std::vector<u64> rows((size_t)want);
        u32 n = 0;
        int rc = sa_hip_index_query_rows(c->idx, reinterpret_cast<const uint8_t*>(substring), strlen(substring), want, rows.data(), &n, nullptr);
        if (rc) return rc;
        for (u32 i = 0; i < n; ++i) {
            const u64 r = rows[i];
            u64 b = c->row_file_offsets[r], e = c->row_file_offsets[r + 1];
            while (e > b && (c->map[e - 1] == '\n' || c->map[e - 1] == '\r')) --e;
            row_ptrs[i] = reinterpret_cast<const char*>(c->map + b);
            row_lens[i] = (u32)std::min<u64>(e - b, 0xFFFFFFFFull);
        }
        *num_matches = n;
    } catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "sa_hip_get_matching_row_spans_file: out of host memory"); }
    return 0;
}

uint32_t sa_hip_get_matching_records(const char* str, const sa_hip_SuffixArray_struct* s, const char* substring, uint32_t k,
                                     char** matching_records) {
    if (!str || !s || !substring || (!matching_records && k) || (!s->suffix_array && s->n)) { fail(SA_HIP_EINVAL, "sa_hip_get_matching_records: NULL argument"); return 0; }
    const sa_hip_pair_u32 rg = sa_hip_get_substring_positions(str, s, substring);
    if (rg.first == 0xFFFFFFFFu || (u32)(rg.second - rg.first + 1u) == 0u) return 0;   // engine.c:1181-1185
    const u64 n = s->n;
    u32 made = 0;
    try {
        std::unordered_set<u64> seen;
        for (u64 i = rg.first; i <= (u64)rg.second && made < k; ++i) {
            const u64 p = s->suffix_array[i];
            // the record = the text between the newlines around the hit (what engine.c:1187-1211 means to copy)
            u64 b = p, e = p;
            while (b > 0 && str[b - 1] != '\n') --b;
            while (e < n && str[e] != '\n') ++e;
            if (!seen.insert(b).second) continue;
            char* r = static_cast<char*>(malloc((size_t)(e - b) + 1));
            if (!r) { fail(SA_HIP_ENOMEM, "sa_hip_get_matching_records: out of host memory"); return made; }
            memcpy(r, str + b, (size_t)(e - b));
            r[e - b] = '\0';
            matching_records[made++] = r;
        }
    } catch (const std::bad_alloc&) { fail(SA_HIP_ENOMEM, "sa_hip_get_matching_records: out of host memory"); }
    return made;
}

void sa_hip_free_records(char** records, uint32_t n) {
    if (!records) return;
    for (uint32_t i = 0; i < n; ++i) { free(records[i]); records[i] = nullptr; }
}

// ---- lifecycle mirrors of the seam (engine.c:326-349; pyx:68-74) -------------------------------------------------------

int sa_hip_init_suffix_array_byte_idxs(sa_hip_SuffixArray_struct* s, uint32_t max_suffix_length, uint64_t global_byte_start_idx,
                                       uint64_t global_byte_end_idx, uint32_t n) {
    if (!s) return fail(SA_HIP_EINVAL, "sa_hip_init_suffix_array_byte_idxs: NULL argument");
    s->max_suffix_length = max_suffix_length;
    s->n = n;
    s->suffix_array = static_cast<uint32_t*>(malloc((size_t)(n ? n : 1) * sizeof(uint32_t)));
    s->is_quoted_bitflag = nullptr;   // the row table of sa_hip_csv_index replaces the reference's per-character quoted bits
    s->global_byte_start_idx = global_byte_start_idx;
    s->global_byte_end_idx = global_byte_end_idx;
    if (!s->suffix_array) return fail(SA_HIP_ENOMEM, "sa_hip_init_suffix_array_byte_idxs: out of host memory");
    return 0;
}

void sa_hip_free_suffix_array(sa_hip_SuffixArray_struct* s) {
    if (!s) return;
    free(s->suffix_array);
    s->suffix_array = nullptr;
}

// ---- the reference's on-disk layout (engine.c:1098-1165) ----------------------------------------------------------------

int sa_hip_write_suffix_array(const sa_hip_SuffixArray_struct* s, const char* sa_filename, const char* is_quoted_filename) {
    if (!s || !sa_filename || (!s->suffix_array && s->n)) return fail(SA_HIP_EINVAL, "sa_hip_write_suffix_array: NULL argument");
    FILE* f = fopen(sa_filename, "wb");
    if (!f) return fail(SA_HIP_EINVAL, "sa_hip_write_suffix_array: cannot open file", sa_filename);
    // engine.c:1123-1128: {u64 global_byte_start_idx, u64 global_byte_end_idx, u32 max_suffix_length, u32 n, u32 suffix_array[n]}
    bool ok = fwrite(&s->global_byte_start_idx, 8, 1, f) == 1 && fwrite(&s->global_byte_end_idx, 8, 1, f) == 1 &&
              fwrite(&s->max_suffix_length, 4, 1, f) == 1 && fwrite(&s->n, 4, 1, f) == 1 &&
              (s->n == 0 || fwrite(s->suffix_array, 4, s->n, f) == s->n);
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(SA_HIP_EINVAL, "sa_hip_write_suffix_array: short write", sa_filename);
    if (is_quoted_filename) {
        // engine.c:1098-1101 write_buffer_bit: {u32 capacity, bytes}; this library keeps no quoted bits (sa_hip.h): capacity 0
        FILE* q = fopen(is_quoted_filename, "wb");
        if (!q) return fail(SA_HIP_EINVAL, "sa_hip_write_suffix_array: cannot open file", is_quoted_filename);
        const uint32_t cap = 0;
        ok = fwrite(&cap, 4, 1, q) == 1;
        ok = (fclose(q) == 0) && ok;
        if (!ok) return fail(SA_HIP_EINVAL, "sa_hip_write_suffix_array: short write", is_quoted_filename);
    }
    return 0;
}

int sa_hip_read_suffix_array(sa_hip_SuffixArray_struct* s, const char* sa_filename) {
    if (!s || !sa_filename) return fail(SA_HIP_EINVAL, "sa_hip_read_suffix_array: NULL argument");
    memset(s, 0, sizeof *s);
    FILE* f = fopen(sa_filename, "rb");
    if (!f) return fail(SA_HIP_EINVAL, "sa_hip_read_suffix_array: cannot open file", sa_filename);
    int rc = 0;
    if (fread(&s->global_byte_start_idx, 8, 1, f) != 1 || fread(&s->global_byte_end_idx, 8, 1, f) != 1 ||
        fread(&s->max_suffix_length, 4, 1, f) != 1 || fread(&s->n, 4, 1, f) != 1)
        rc = fail(SA_HIP_EINVAL, "sa_hip_read_suffix_array: truncated header", sa_filename);
    if (!rc) {
        s->suffix_array = static_cast<uint32_t*>(malloc((size_t)(s->n ? s->n : 1) * 4));
        if (!s->suffix_array) rc = fail(SA_HIP_ENOMEM, "sa_hip_read_suffix_array: out of host memory");
        else if (s->n && fread(s->suffix_array, 4, s->n, f) != s->n) rc = fail(SA_HIP_EINVAL, "sa_hip_read_suffix_array: truncated array", sa_filename);
        if (!rc) for (uint32_t i = 0; i < s->n; ++i) if (s->suffix_array[i] >= s->n) { rc = fail(SA_HIP_EINVAL, "sa_hip_read_suffix_array: entry >= n", sa_filename); break; }
    }
    fclose(f);
    if (rc) { free(s->suffix_array); memset(s, 0, sizeof *s); }
    return rc;
}

// ---- libsais-call-compatible wrappers --------------------------------------------------------------
// Host pointers in, host suffix array out.  One process-level workspace serves all of them (host_io.hpp): a cached
// index handle whose device buffers are allocated once and grow on demand, and a ring of pinned slabs for both legs
// over PCIe.  sa_hip_release_workspace() gives the memory back; sa_hip_last_call_breakdown() tells where the time of the
// last call went.

extern "C++" {
namespace {
double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// OUT = uint32_t / int32_t (libsais layout) or int64_t (libsais64 layout)
template <typename OUT, typename FREQ>
int oneshot_build(const uint8_t* T, uint64_t n, uint32_t L, OUT* out, FREQ* freq) {
    OneShot& g = g_oneshot;
    std::lock_guard<std::mutex> lock(g.mu);
    const auto t_all = std::chrono::steady_clock::now();
    if (!g.idx && host_path_allowed() && sa_hip_device_count() <= 0) {   // opt-in no-GPU path (host_index.hpp)
        if (n > HOST_MAX_N) return fail(SA_HIP_EINVAL, "the host path (no HIP device) holds at most 2^24 bytes");
        try {
            HostIndex h;
            h.set_text(T, n);
            h.build(L);
            for (u64 i = 0; i < n; ++i) out[i] = (OUT)h.sa[i];
            if (freq) for (int c = 0; c < 256; ++c) freq[c] = (FREQ)h.freq[c];
        } catch (const std::bad_alloc&) { return fail(SA_HIP_ENOMEM, "out of host memory"); }
        return 0;
    }
    sa_hip_call_breakdown bd{};
    bd.n = n;
    int rc = 0;
    // workspace: reuse the cached handle when it is large enough
    auto t0 = std::chrono::steady_clock::now();
    bd.workspace_reused = (g.idx && g.idx->b.n_max >= n && g.ring.ready) ? 1u : 0u;
    if (g.idx && g.idx->b.n_max < n) { sa_hip_index_destroy(g.idx); g.idx = nullptr; }
    if (!g.idx && (rc = sa_hip_index_create(&g.idx, n ? n : 1, 0))) return rc;
    sa_hip_index* idx = g.idx;
    std::lock_guard<std::mutex> ilock(idx->mu);
    if ((rc = set_device(idx->device))) return rc;
    if (g.ring.ready && g.ring.device != idx->device) g.ring.destroy();   // (its copy stream belongs to another device: a rows batch made it there)
    if ((rc = g.ring.init())) return rc;
    bd.workspace_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if ((rc = ring_upload(g.ring, idx->stream, idx->device, idx->b.text.p, T, (size_t)n))) return rc;
    bd.upload_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    idx->has_index = false;
    idx->widen_ms = 0.0;
    rc = idx->b.build(n, L);
    idx->has_index = (rc == 0);
    if (rc) return rc;
    SA_HIP_CHECK(hipStreamSynchronize(idx->stream));
    bd.build_ms = ms_since(t0);
    bd.build_device_ms = idx->b.stats.total_ms;
    t0 = std::chrono::steady_clock::now();
    if ((rc = ring_download<OUT>(g.ring, idx->device, idx->b.sa, out, (size_t)n))) return rc;
    bd.download_ms = ms_since(t0);
    if (freq) for (int c = 0; c < 256; ++c) freq[c] = (FREQ)idx->b.freq[c];
    bd.total_ms = ms_since(t_all);
    g.last = bd;
    return 0;
}
}  // namespace
}  // extern "C++"

int sa_hip_last_call_breakdown(sa_hip_call_breakdown* out) {
    if (!out) return fail(SA_HIP_EINVAL, "sa_hip_last_call_breakdown: NULL argument");
    std::lock_guard<std::mutex> lock(g_oneshot.mu);
    *out = g_oneshot.last;
    return 0;
}

void sa_hip_release_workspace(void) {
    std::lock_guard<std::mutex> lock(g_oneshot.mu);
    if (g_oneshot.idx) { (void)hipSetDevice(g_oneshot.idx->device); sa_hip_index_destroy(g_oneshot.idx); g_oneshot.idx = nullptr; }
    g_oneshot.ring.destroy();
}

int32_t sa_hip_libsais_omp(const uint8_t* T, int32_t* SA, int32_t n, int32_t fs, int32_t* freq, int32_t threads) {
    if (T == nullptr || SA == nullptr || n < 0 || fs < 0 || threads < 0) return fail(SA_HIP_EINVAL, "sa_hip_libsais: invalid arguments");
    return oneshot_build<int32_t, int32_t>(T, (uint64_t)n, 0, SA, freq);
}

int32_t sa_hip_libsais(const uint8_t* T, int32_t* SA, int32_t n, int32_t fs, int32_t* freq) {
    return sa_hip_libsais_omp(T, SA, n, fs, freq, 0);
}

// ---- texts beyond 2^32 - 2 bytes: 64-bit suffix indices (big_build.hpp; libsais64.c:6684 -> libsais64_main) ----------------
int sa_hip_libsais64_device(const void* text_dev, int64_t* sa_dev, int64_t n, int device, sa_hip_big_stats* stats_out) {
    if ((!text_dev || !sa_dev) && n) return fail(SA_HIP_EINVAL, "sa_hip_libsais64_device: NULL argument");
    if (n < 0) return fail(SA_HIP_EINVAL, "sa_hip_libsais64_device: negative length");
    int rc = set_device(device);
    if (rc) return rc;
    big::BigBuilder b;
    SA_HIP_CHECK(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
    rc = b.build(static_cast<const u8*>(text_dev), (u64)n, reinterpret_cast<u64*>(sa_dev));
    (void)hipStreamSynchronize(b.stream);
    if (stats_out) {
        stats_out->sigma = b.stats.sigma; stats_out->bits_per_symbol = b.stats.bits_per_symbol; stats_out->initial_chars = b.stats.initial_chars;
        stats_out->sort_passes = b.stats.sort_passes; stats_out->rounds = b.stats.rounds; stats_out->pad_ = 0;
        stats_out->tied_after_sort = b.stats.tied_after_sort; stats_out->tied_total = b.stats.tied_total; stats_out->total_ms = b.stats.total_ms;
    }
    b.destroy();
    (void)hipStreamDestroy(b.stream);
    return rc;
}

int sa_hip_sufcheck64_device(const void* text_dev, const int64_t* sa_dev, int64_t n, int device, uint64_t* violations) {
    if (!violations || ((!text_dev || !sa_dev) && n) || n < 0) return fail(SA_HIP_EINVAL, "sa_hip_sufcheck64_device: invalid arguments");
    int rc = set_device(device);
    if (rc) return rc;
    big::BigBuilder b;
    SA_HIP_CHECK(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
    u64 v = 0;
    rc = b.verify(static_cast<const u8*>(text_dev), reinterpret_cast<const u64*>(sa_dev), (u64)n, &v);
    *violations = v;
    b.destroy();
    (void)hipStreamDestroy(b.stream);
    return rc;
}

namespace {
// host text in, host suffix array out, through plain device buffers (a call of this size is dominated by its 8 n bytes over PCIe)
int big_oneshot(const uint8_t* T, uint64_t n, int64_t* SA, int64_t* freq) {
    int rc = set_device(0);
    if (rc) return rc;
    DevBuf text, sa;
    if ((rc = text.ensure(n + 64)) || (rc = sa.ensure(n * 8 + 64))) { text.release(); sa.release(); return rc; }
    auto body = [&]() -> int {
        SA_HIP_CHECK(hipMemcpy(text.p, T, n, hipMemcpyHostToDevice));
        int r = sa_hip_libsais64_device(text.p, sa.as<int64_t>(), (int64_t)n, 0, nullptr);
        if (r) return r;
        SA_HIP_CHECK(hipMemcpy(SA, sa.p, n * 8, hipMemcpyDeviceToHost));
        return 0;
    };
    rc = body();
    text.release(); sa.release();
    if (rc == 0 && freq) {   // (libsais.h:84: the byte histogram, when asked for)
        for (int c = 0; c < 256; ++c) freq[c] = 0;
        for (uint64_t i = 0; i < n; ++i) ++freq[T[i]];
    }
    return rc;
}
}  // namespace

int64_t sa_hip_libsais64_omp(const uint8_t* T, int64_t* SA, int64_t n, int64_t fs, int64_t* freq, int64_t threads) {
    if (T == nullptr || SA == nullptr || n < 0 || fs < 0 || threads < 0) return fail(SA_HIP_EINVAL, "sa_hip_libsais64: invalid arguments");
    if ((uint64_t)n > 0xFFFFFFFEull) return big_oneshot(T, (uint64_t)n, SA, freq);   // 64-bit suffix indices (big_build.hpp)
    return oneshot_build<int64_t, int64_t>(T, (uint64_t)n, 0, SA, freq);
}

int64_t sa_hip_libsais64(const uint8_t* T, int64_t* SA, int64_t n, int64_t fs, int64_t* freq) {
    return sa_hip_libsais64_omp(T, SA, n, fs, freq, 0);
}

// ---- engine.c-call-compatible wrappers ---------------------------------------------------------------

int sa_hip_construct_truncated_suffix_array(const char* text, sa_hip_SuffixArray_struct* s) {
    if (!text || !s || (!s->suffix_array && s->n)) return fail(SA_HIP_EINVAL, "sa_hip_construct_truncated_suffix_array: NULL argument");
    // engine.c:841: depth = min(max_suffix_length, n); 0 would mean "no order at all" there,
    // which the device build expresses as L >= 1 only, so L == 0 degenerates to identity order.
    if (s->max_suffix_length == 0) {
        for (uint32_t i = 0; i < s->n; ++i) s->suffix_array[i] = i;
        return 0;
    }
    return oneshot_build<uint32_t, uint64_t>(reinterpret_cast<const uint8_t*>(text), s->n, s->max_suffix_length, s->suffix_array, nullptr);
}

sa_hip_pair_u32 sa_hip_get_substring_positions(const char* str, const sa_hip_SuffixArray_struct* s, const char* substring) {
    sa_hip_pair_u32 r = {0xFFFFFFFFu, 0xFFFFFFFFu};
    if (!str || !s || !substring || (!s->suffix_array && s->n)) { fail(SA_HIP_EINVAL, "sa_hip_get_substring_positions: NULL argument"); return r; }
    TempIndex t;
    if (sa_hip_index_create(&t.idx, s->n, 0)) return r;
    // max_suffix_length == 0 in the reference means cmp_length 0 (everything matches); the handle
    // API uses 0 for "unlimited", so pass the pattern truncated accordingly.
    const uint64_t m = strlen(substring);
    const uint32_t L = s->max_suffix_length;
    if (sa_hip_index_load(t.idx, reinterpret_cast<const uint8_t*>(str), s->suffix_array, s->n, L ? L : 1)) return r;
    const uint64_t off[2] = {0, L ? m : 0};
    if (sa_hip_query_batch(t.idx, reinterpret_cast<const uint8_t*>(substring), off, 1, &r)) {
        r.first = r.second = 0xFFFFFFFFu;
    }
    return r;
}

int sa_hip_sort_pairs(uint64_t* keys, uint32_t* values, uint64_t n, int begin_bit, int end_bit, int device) {
    if ((!keys && n) || begin_bit < 0 || end_bit > 64 || begin_bit > end_bit || n > 0xFFFFFFFEull)
        return fail(SA_HIP_EINVAL, "sa_hip_sort_pairs: invalid arguments");
    if (n == 0 || begin_bit == end_bit) return 0;
    int rc = set_device(device);
    if (rc) return rc;
    hipStream_t stream;
    SA_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    RadixWorkspace ws;
    DevBuf k0, k1, v0, v1;
    int sort_block = 512;
    if (const char* e = diag_env("SA_HIP_SORT_BLOCK")) sort_block = (atoi(e) == 256) ? 256 : 512;
    rc = ws.init(n, sort_block);
    if (!rc) rc = k0.ensure(n * 8);
    if (!rc) rc = k1.ensure(n * 8);
    if (!rc) rc = v0.ensure(n * 4);
    if (!rc) rc = v1.ensure(n * 4);
    u64* kr = nullptr; u32* vr = nullptr;
    auto body = [&]() -> int {
        SA_HIP_CHECK(hipMemcpyAsync(k0.p, keys, n * 8, hipMemcpyHostToDevice, stream));
        if (values) SA_HIP_CHECK(hipMemcpyAsync(v0.p, values, n * 4, hipMemcpyHostToDevice, stream));
        int r = radix_sort_pairs(ws, stream, k0.as<u64>(), v0.as<u32>(), k1.as<u64>(), v1.as<u32>(), (u32)n, begin_bit,
                                 end_bit, values == nullptr, false, &kr, &vr);
        if (r) return r;
        SA_HIP_CHECK(hipMemcpyAsync(keys, kr, n * 8, hipMemcpyDeviceToHost, stream));
        if (values) SA_HIP_CHECK(hipMemcpyAsync(values, vr, n * 4, hipMemcpyDeviceToHost, stream));
        DeviceStatus st;
        SA_HIP_CHECK(hipMemcpyAsync(&st, ws.dstat, sizeof st, hipMemcpyDeviceToHost, stream));
        SA_HIP_CHECK(hipStreamSynchronize(stream));
        if (st.error) return fail(SA_HIP_EINTERNAL, "device look-back spin limit expired");
        return ws.timer.flush();
    };
    if (!rc) rc = body();
    (void)hipStreamSynchronize(stream);
    k0.release(); k1.release(); v0.release(); v1.release();
    ws.destroy();
    (void)hipStreamDestroy(stream);
    return rc;
}

int sa_hip_csv_extract_column(const char* path, const char* column, sa_hip_csv_column* out) {
    if (!path || !column || !out) return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column: NULL argument");
    // nothing may unwind through the C ABI: allocation failures and thread-creation failures become codes
    try {
        return csv_extract_column(path, column, out);
    } catch (const std::bad_alloc&) {
        csv_free(out);
        return fail(SA_HIP_ENOMEM, "sa_hip_csv_extract_column: out of host memory");
    } catch (const std::exception& e) {
        csv_free(out);
        return fail(SA_HIP_EINVAL, "sa_hip_csv_extract_column", e.what());
    }
}
void sa_hip_csv_free(sa_hip_csv_column* col) { csv_free(col); }
int sa_hip_synth_csv(const char* path, uint64_t rows, uint64_t seed) {
    if (!path) return fail(SA_HIP_EINVAL, "sa_hip_synth_csv: NULL path");
    try {
        return synth_csv(path, rows, seed);
    } catch (const std::bad_alloc&) {
        return fail(SA_HIP_ENOMEM, "sa_hip_synth_csv: out of host memory");
    } catch (const std::exception& e) {
        return fail(SA_HIP_EINVAL, "sa_hip_synth_csv", e.what());
    }
}

void sa_hip_synth_uniform27(uint8_t* out, uint64_t n, uint64_t seed) {
    // xorshift64 (13,7,17), symbol = (s >> 33) % 27, 26 -> '\n'
    uint64_t s = seed ? seed : 88172645463325252ull;
    for (uint64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t v = (uint32_t)((s >> 33) % 27u);
        out[i] = (uint8_t)(v == 26 ? '\n' : 'a' + v);
    }
}

}  // extern "C"
