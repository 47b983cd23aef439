// host_io.hpp -- the host <-> device legs of the libsais-call-compatible wrappers (sa_hip_libsais[64][_omp],
// sa_hip_construct_truncated_suffix_array): host pointers in, host suffix array out (libsais.h:84, libsais64.h:61).
//
// The device build of 1e9 characters takes ~21 ms; a caller that hands over pageable host memory used to wait ~1 s:
// ~18 GB of device buffers allocated and freed per call, the text copied from pageable memory, and the int64 result
// widened on the device and copied slab by slab, strictly serially, 8 GB over PCIe.  Here
//   * the one-shot wrappers share ONE process-level workspace (a cached index handle + a ring of pinned slabs),
//   * the text goes up through the ring (worker threads copy into pinned slabs while the previous slab's DMA runs),
//   * the result comes down as u32 -- 4 bytes per entry over PCIe instead of 8 -- into pinned slabs on a second stream
//     while worker threads widen (libsais64 layout, libsais64.c:6248-6259) or copy the slabs that have arrived into
//     the caller's array: PCIe, the host's memory bandwidth and the caller's first-touch page faults overlap.
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include "common.hpp"

namespace sa {

inline unsigned host_workers(unsigned cap = 16) {
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const unsigned a = (unsigned)CPU_COUNT(&set); if (a && a < hw) hw = a; }
    if (hw < 1) hw = 1;
    return hw < cap ? hw : cap;
}

struct PinnedRing {
    static constexpr int SLABS = 16;
    static constexpr size_t SLAB_BYTES = 32u << 20;   // 16 x 32 MiB pinned
    u8* slab[SLABS] = {};
    hipEvent_t ev[SLABS] = {};
    hipStream_t copy_stream = nullptr;                 // the down leg runs beside the index's own stream
    bool ready = false;
    int device = -1;                                   // the device that was current when copy_stream was created
    int init() {
        if (ready) return 0;
        SA_HIP_CHECK(hipGetDevice(&device));
        for (int i = 0; i < SLABS; ++i) {   // (a call that failed half way is completed by the next one: nothing is allocated twice)
            if (!slab[i]) SA_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&slab[i]), SLAB_BYTES, hipHostMallocDefault));
            if (!ev[i]) SA_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        if (!copy_stream) SA_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        ready = true;
        return 0;
    }
    void destroy() {
        for (int i = 0; i < SLABS; ++i) {
            if (slab[i]) (void)hipHostFree(slab[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            slab[i] = nullptr; ev[i] = nullptr;
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        copy_stream = nullptr;
        ready = false;
    }
};

// src[0..bytes) (pageable host memory) -> dst_dev on `stream`: worker w fills slab k = w, w + W, ... ; the DMA of a slab
// is issued by the worker that filled it (HIP calls are thread-safe; order between slabs does not matter)
inline int ring_upload(PinnedRing& r, hipStream_t stream, int device, void* dst_dev, const u8* src, size_t bytes) {
    if (bytes == 0) return 0;
    const size_t nslab = (bytes + PinnedRing::SLAB_BYTES - 1) / PinnedRing::SLAB_BYTES;
    // worker w owns the slabs {w, w + W, ...} mod SLABS: W divides SLABS, so no two workers ever meet on a slab
    unsigned W = (unsigned)std::min<size_t>(std::min<size_t>(host_workers(), PinnedRing::SLABS), nslab);
    while (PinnedRing::SLABS % W) --W;
    std::atomic<int> err{0};
    std::vector<std::thread> th;
    auto work_d = [&](unsigned w) {
        if (hipSetDevice(device) != hipSuccess) { err = 1; return; }
        for (size_t k = w; k < nslab && !err; k += W) {
            const int s = (int)(k % PinnedRing::SLABS);
            const size_t off = k * PinnedRing::SLAB_BYTES;
            const size_t len = std::min(PinnedRing::SLAB_BYTES, bytes - off);
            if (k >= (size_t)PinnedRing::SLABS && hipEventSynchronize(r.ev[s]) != hipSuccess) { err = 1; return; }   // the slab's previous DMA
            memcpy(r.slab[s], src + off, len);
            if (hipMemcpyAsync(static_cast<u8*>(dst_dev) + off, r.slab[s], len, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipEventRecord(r.ev[s], stream) != hipSuccess) { err = 1; return; }
        }
    };
    try {
        for (unsigned w = 1; w < W; ++w) th.emplace_back(work_d, w);
    } catch (...) { err = 1; }
    work_d(0);
    for (auto& t : th) t.join();
    if (err) return fail(SA_HIP_EHIP, "ring_upload: host-to-device copy failed");
    SA_HIP_CHECK(hipStreamSynchronize(stream));
    return 0;
}

// sa_dev[0..n) (u32, device) -> out[0..n) on the host as OUT (u32 / int32: copy; int64: widened).  The DMA of slab k
// runs on the ring's copy stream while workers convert the slabs that have arrived.  The caller has synchronised
// the stream that produced sa_dev.
template <typename OUT>
inline int ring_download(PinnedRing& r, int device, const u32* sa_dev, OUT* out, size_t n) {
    if (n == 0) return 0;
    constexpr size_t PER = PinnedRing::SLAB_BYTES / 4;   // entries per slab
    const size_t nslab = (n + PER - 1) / PER;
    unsigned W = (unsigned)std::min<size_t>(std::min<size_t>(host_workers(), PinnedRing::SLABS), nslab);
    while (PinnedRing::SLABS % W) --W;
    std::atomic<int> err{0};
    auto issue = [&](size_t k) -> bool {
        const int s = (int)(k % PinnedRing::SLABS);
        const size_t off = k * PER;
        const size_t len = std::min(PER, n - off);
        return hipMemcpyAsync(r.slab[s], sa_dev + off, len * 4, hipMemcpyDeviceToHost, r.copy_stream) == hipSuccess &&
               hipEventRecord(r.ev[s], r.copy_stream) == hipSuccess;
    };
    // the first SLABS copies are issued in order up front; afterwards the worker that has emptied a slab issues the copy
    // that refills it
    for (size_t k = 0; k < nslab && k < (size_t)PinnedRing::SLABS; ++k)
        if (!issue(k)) return fail(SA_HIP_EHIP, "ring_download: device-to-host copy failed");
    auto work = [&](unsigned w) {
        if (hipSetDevice(device) != hipSuccess) { err = 1; return; }
        for (size_t k = w; k < nslab && !err; k += W) {
            const int s = (int)(k % PinnedRing::SLABS);
            const size_t off = k * PER;
            const size_t len = std::min(PER, n - off);
            if (hipEventSynchronize(r.ev[s]) != hipSuccess) { err = 1; return; }
            const u32* in = reinterpret_cast<const u32*>(r.slab[s]);
            OUT* o = out + off;
            if (sizeof(OUT) == 4) memcpy(o, in, len * 4);
            else for (size_t i = 0; i < len; ++i) o[i] = (OUT)in[i];
            if (k + PinnedRing::SLABS < nslab && !issue(k + PinnedRing::SLABS)) { err = 1; return; }
        }
    };
    std::vector<std::thread> th;
    try {
        for (unsigned w = 1; w < W; ++w) th.emplace_back(work, w);
    } catch (...) { err = 1; }
    work(0);
    for (auto& t : th) t.join();
    if (err) return fail(SA_HIP_EHIP, "ring_download: device-to-host copy failed");
    return 0;
}

// src_dev[0..bytes) -> consume(piece, byte offset, byte length) for consecutive pieces of `piece_bytes` (<= SLAB_BYTES; the
// caller picks a multiple of its record size so that no record straddles two pieces).  Same pipeline as ring_download: the
// DMA of piece k + SLABS is issued by the worker that has consumed piece k; consume() runs on up to 16 worker threads at once,
// on disjoint pieces, and must only write what belongs to its piece.  The caller has synchronised the stream that produced
// src_dev.  (The batched record retrieval, sa_hip_index_query_rows_batch: Q x k row ids come down as u32 and are widened
// into the caller's uint64[Q][k] -- its first-touch page faults spread over the workers instead of one thread.)
template <typename FN>
inline int ring_download_pieces(PinnedRing& r, int device, const u8* src_dev, size_t bytes, size_t piece_bytes, FN consume) {
    if (bytes == 0) return 0;
    if (piece_bytes == 0 || piece_bytes > PinnedRing::SLAB_BYTES) return fail(SA_HIP_EINVAL, "ring_download_pieces: piece size");
    const size_t npiece = (bytes + piece_bytes - 1) / piece_bytes;
    unsigned W = (unsigned)std::min<size_t>(std::min<size_t>(host_workers(), PinnedRing::SLABS), npiece);
    while (PinnedRing::SLABS % W) --W;
    std::atomic<int> err{0};
    auto issue = [&](size_t k) -> bool {
        const int s = (int)(k % PinnedRing::SLABS);
        const size_t off = k * piece_bytes;
        const size_t len = std::min(piece_bytes, bytes - off);
        return hipMemcpyAsync(r.slab[s], src_dev + off, len, hipMemcpyDeviceToHost, r.copy_stream) == hipSuccess &&
               hipEventRecord(r.ev[s], r.copy_stream) == hipSuccess;
    };
    for (size_t k = 0; k < npiece && k < (size_t)PinnedRing::SLABS; ++k)
        if (!issue(k)) return fail(SA_HIP_EHIP, "ring_download_pieces: device-to-host copy failed");
    auto work = [&](unsigned w) {
        if (hipSetDevice(device) != hipSuccess) { err = 1; return; }
        for (size_t k = w; k < npiece && !err; k += W) {
            const int s = (int)(k % PinnedRing::SLABS);
            const size_t off = k * piece_bytes;
            const size_t len = std::min(piece_bytes, bytes - off);
            if (hipEventSynchronize(r.ev[s]) != hipSuccess) { err = 1; return; }
            consume(static_cast<const u8*>(r.slab[s]), off, len);
            if (k + PinnedRing::SLABS < npiece && !issue(k + PinnedRing::SLABS)) { err = 1; return; }
        }
    };
    std::vector<std::thread> th;
    try {
        for (unsigned w = 1; w < W; ++w) th.emplace_back(work, w);
    } catch (...) { err = 1; }
    work(0);
    for (auto& t : th) t.join();
    if (err) return fail(SA_HIP_EHIP, "ring_download_pieces: device-to-host copy failed");
    return 0;
}

}  // namespace sa
