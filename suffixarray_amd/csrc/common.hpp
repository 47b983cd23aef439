// common.hpp -- shared host/device helpers for libsa_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <new>

#include "../../include/sa_hip.h"

namespace sa {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;

constexpr int WAVE = 64;  // gfx950 wavefront width

// thread-local last error text (sa_hip_last_error)
inline std::string& last_error() {
    static thread_local std::string s;
    return s;
}
inline int fail(int code, const char* what, const char* detail = nullptr) {
    std::string& e = last_error();
    e = what;
    if (detail) { e += ": "; e += detail; }
    return code;
}

#define SA_HIP_CHECK(expr)                                                                 \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            char _b[256];                                                                  \
            snprintf(_b, sizeof _b, "%s:%d %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return ::sa::fail(_e == hipErrorOutOfMemory ? SA_HIP_ENOMEM : SA_HIP_EHIP, #expr, _b); \
        }                                                                                  \
    } while (0)

__host__ __device__ inline u32 div_up(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// smallest b with 2^b >= count
inline int bits_for(u64 count) {
    int b = 0;
    while (b < 64 && (1ull << b) < count) ++b;
    return b;
}

#if defined(__HIPCC__)   // device helpers: absent when a host compiler builds the host-only headers (tools/host_sanitize.cpp)
__device__ __forceinline__ u64 lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// Workgroup barrier that first drains this wave's outstanding LDS operations explicitly.
// hipcc (ROCm 7.2, gfx950) was observed to emit a bare `s_barrier` -- no `s_waitcnt lgkmcnt(0)` --
// on a loop path where the only pending LDS operations were no-return atomics (`ds_add_u32`)
// issued in the previous iteration: another wave then read the counter before the add landed and
// a whole wave-instruction of increments was lost (found on 6.7e7-record sorts: a per-chunk
// histogram short by exactly 64).  Every barrier that follows LDS atomics uses this form; inline
// asm is invisible to the waitcnt-insertion pass, so the wait cannot be dropped.
__device__ __forceinline__ void sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
#endif

// A u64 table on the host with std::vector's read interface that can also ADOPT a malloc'ed array (the CSV extractor's
// row tables: 2 x 400 MB at 50M rows -- copying them into vectors cost 0.2 s of an index build that takes 0.06 s on the device).
struct HostU64Array {
    u64* p = nullptr;
    size_t n = 0;
    HostU64Array() = default;
    HostU64Array(const HostU64Array&) = delete;
    HostU64Array& operator=(const HostU64Array&) = delete;
    ~HostU64Array() { free(p); }
    void clear() { free(p); p = nullptr; n = 0; }
    void adopt(u64* malloced, size_t count) { clear(); p = malloced; n = count; }      // takes ownership (free())
    void assign(const u64* first, const u64* last) {                                   // copies; throws std::bad_alloc
        const size_t count = (size_t)(last - first);
        u64* q = static_cast<u64*>(malloc((count ? count : 1) * sizeof(u64)));
        if (!q) throw std::bad_alloc();
        if (count) memcpy(q, first, count * sizeof(u64));
        clear(); p = q; n = count;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const u64* data() const { return p; }
    const u64* begin() const { return p; }
    const u64* end() const { return p + n; }
    const u64& operator[](size_t i) const { return p[i]; }
    const u64& back() const { return p[n - 1]; }
};

// Diagnostic switches (SA_HIP_*): read from the environment ONLY when SA_HIP_DIAG=1 is set as well -- the behaviour of a
// production process does not depend on stray variables of its caller; the tests and tools/ set SA_HIP_DIAG=1 to run
// both plans of a build on the same input.
inline const char* diag_env(const char* name) {
    static const bool on = [] { const char* d = getenv("SA_HIP_DIAG"); return d && atoi(d) != 0; }();
    return on ? getenv(name) : nullptr;
}

// Alphabet compaction of a text: code = 1 + rank of the byte among the bytes that occur; 0 = past the end.
struct CodeMap {
    u16 code[256];  // 0 = byte absent (never looked up for a present position); 1..sigma
};

// Device error word shared by every kernel that can spin (decoupled look-back).
// 0 = ok; anything else = a bounded spin expired, results are invalid.
struct DeviceStatus {
    u32 error;
    u32 pad[3];
};

}  // namespace sa
