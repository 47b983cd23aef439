// sortbench.hip -- ablation micro-benchmark of ONE radix pass (diagnostic build, not shipped).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/sortbench tools/sortbench.hip && /tmp/sortbench [log2n]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../suffixarray_amd/csrc/radix_sort.hpp"
using namespace sa;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill_keys(u64* k, u32* v, u64 n, int bits) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        u64 x = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
        k[i] = x << (64 - bits);
        v[i] = (u32)i;
    }
}
__global__ void copy_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, u64 n16) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) b[i] = a[i];
}

__global__ void copy8_kernel(const u64* __restrict__ a, u64* __restrict__ b, u64 n) {   // 8 B per lane: the key access width
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
__global__ void copy4_kernel(const u32* __restrict__ a, u32* __restrict__ b, u64 n) {   // 4 B per lane: the value access width
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}

// copy variants: what does the chip stream at, for the access shapes of the sort?
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy16u_kernel(const uint4* __restrict__ a_, uint4* __restrict__ b_, u64 n16) {
    const v4u* __restrict__ a = reinterpret_cast<const v4u*>(a_);
    v4u* __restrict__ b = reinterpret_cast<v4u*>(b_);
    // each block owns a contiguous span of 256 * UNROLL uint4 per iteration
    const u64 span = (u64)256 * UNROLL;
    for (u64 base = (u64)blockIdx.x * span; base < n16; base += (u64)gridDim.x * span) {
        v4u r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u64 i = base + (u64)u * 256 + threadIdx.x;
            if (i < n16) r[u] = NT ? __builtin_nontemporal_load(&a[i]) : a[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u64 i = base + (u64)u * 256 + threadIdx.x;
            if (i < n16) { if (NT) __builtin_nontemporal_store(r[u], &b[i]); else b[i] = r[u]; }
        }
    }
}
// the sort's shape: 512 threads, one 8192-record tile per block, 8-byte key and 4-byte value accesses
__global__ __launch_bounds__(512) void copy_tile_kernel(const u64* __restrict__ k, const u32* __restrict__ v, u64* __restrict__ ko,
                                                        u32* __restrict__ vo, u32 n) {
    const u64 base = (u64)blockIdx.x * 8192;
    u64 kk[16]; u32 vv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { const u64 i = base + j * 512 + threadIdx.x; kk[j] = i < n ? k[i] : 0; }
#pragma unroll
    for (int j = 0; j < 16; ++j) { const u64 i = base + j * 512 + threadIdx.x; vv[j] = i < n ? v[i] : 0; }
#pragma unroll
    for (int j = 0; j < 16; ++j) { const u64 i = base + j * 512 + threadIdx.x; if (i < n) ko[i] = kk[j]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) { const u64 i = base + j * 512 + threadIdx.x; if (i < n) vo[i] = vv[j]; }
}

// narrow-record shapes: (u32 key, u32 value) tiles; LD16: 16-byte loads (4 records per lane), ST16: 16-byte stores
template <bool LD16, bool ST16>
__global__ __launch_bounds__(512) void copy_tile32_kernel(const u32* __restrict__ k, const u32* __restrict__ v, u32* __restrict__ ko,
                                                          u32* __restrict__ vo, u32 n) {
    const u64 base = (u64)blockIdx.x * 8192;
    if (base + 8192 > n) return;   // full tiles only (timing)
    u32 kk[16], vv[16];
    if (LD16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint4 x = *reinterpret_cast<const uint4*>(k + base + (j * 512 + threadIdx.x) * 4); kk[4*j] = x.x; kk[4*j+1] = x.y; kk[4*j+2] = x.z; kk[4*j+3] = x.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint4 x = *reinterpret_cast<const uint4*>(v + base + (j * 512 + threadIdx.x) * 4); vv[4*j] = x.x; vv[4*j+1] = x.y; vv[4*j+2] = x.z; vv[4*j+3] = x.w; }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) kk[j] = k[base + j * 512 + threadIdx.x];
#pragma unroll
        for (int j = 0; j < 16; ++j) vv[j] = v[base + j * 512 + threadIdx.x];
    }
    if (ST16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(ko + base + (j * 512 + threadIdx.x) * 4) = make_uint4(kk[4*j], kk[4*j+1], kk[4*j+2], kk[4*j+3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(vo + base + (j * 512 + threadIdx.x) * 4) = make_uint4(vv[4*j], vv[4*j+1], vv[4*j+2], vv[4*j+3]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) ko[base + j * 512 + threadIdx.x] = kk[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) vo[base + j * 512 + threadIdx.x] = vv[j];
    }
}

template <int BLOCK, int ABL>
float run_pass(RadixWorkspace& ws, hipStream_t st, u64* k0, u32* v0, u64* k1, u32* v1, u32 n, int shift, int reps, int home_mode = 0, u32 incl_mask = SA_INCL_MASK) {
    SortGeom g = make_geom(n, BLOCK * SORT_ITEMS);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(ws.small, 0, RadixWorkspace::zero_bytes(), st));
        u32 hgrid = g.tiles < 2048u ? g.tiles : 2048u;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(hgrid), dim3(256), 0, st, k0, g, shift, 255u, ws.hist(0));
        hipLaunchKernelGGL(radix_scan_hist_kernel, dim3(1), dim3(256), 0, st, ws.hist(0), ws.base());
        SortPassArgs a;
        a.keys_in = k0; a.vals_in = v0; a.keys_out = k1; a.vals_out = v1; a.g = g; a.shift = shift; a.mask = 255u;
        a.next_shift = shift + 8; a.next_mask = 255u; a.next_hist = ws.hist(1);
        a.digit_base = ws.base(); a.status = ws.status; a.ticket = ws.tickets(); a.epoch = ++ws.epoch; a.dstat = ws.dstat; a.home_mode = home_mode; a.incl_mask = incl_mask; a.keys_out32 = nullptr; a.narrow_shift = 0;
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((radix_onesweep_kernel<BLOCK, ABL>), dim3(g.tiles), dim3(BLOCK), 0, st, a);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const u32 n = lg > 64 ? (u32)atoll(argv[1]) : (1u << lg);   // small arg = log2, large arg = record count
    hipStream_t st; CK(hipStreamCreate(&st));
    u64 *k0, *k1; u32 *v0, *v1;
    CK(hipMalloc(&k0, (size_t)n * 8)); CK(hipMalloc(&k1, (size_t)n * 8));
    CK(hipMalloc(&v0, (size_t)n * 4)); CK(hipMalloc(&v1, (size_t)n * 4));
    RadixWorkspace ws;
    if (ws.init(n, 256)) { printf("ws init failed\n"); return 1; }
    fill_keys<<<2048, 256, 0, st>>>(k0, v0, n, 40);
    CK(hipStreamSynchronize(st));
    const double gb = (double)n * 24.0 / 1e9;
    // streaming copy baseline: same bytes (12 B/record in, 12 out)
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, st));
            copy_kernel<<<4096, 256, 0, st>>>((const uint4*)k0, (uint4*)k1, (u64)n * 8 / 16);
            copy_kernel<<<4096, 256, 0, st>>>((const uint4*)v0, (uint4*)v1, (u64)n * 4 / 16);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("n=2^%d  copy(12B/rec)            %8.3f ms  %7.1f GB/s\n", lg, best, gb / best * 1e3);
    }
#define COPYV(name, ...) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float best = 1e9f; \
        for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0, st)); __VA_ARGS__; CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
        printf("copy %-36s %8.3f ms  %7.1f GB/s\n", name, best, gb / best * 1e3); fflush(stdout); }
    {
        const u64 nk = (u64)n * 8 / 16, nv = (u64)n * 4 / 16;
        COPYV("uint4 x1, grid 4096", (copy16u_kernel<1, false><<<4096, 256, 0, st>>>((const uint4*)k0, (uint4*)k1, nk), copy16u_kernel<1, false><<<4096, 256, 0, st>>>((const uint4*)v0, (uint4*)v1, nv)))
        COPYV("uint4 x4, grid 2048", (copy16u_kernel<4, false><<<2048, 256, 0, st>>>((const uint4*)k0, (uint4*)k1, nk), copy16u_kernel<4, false><<<2048, 256, 0, st>>>((const uint4*)v0, (uint4*)v1, nv)))
        COPYV("uint4 x8, grid 2048", (copy16u_kernel<8, false><<<2048, 256, 0, st>>>((const uint4*)k0, (uint4*)k1, nk), copy16u_kernel<8, false><<<2048, 256, 0, st>>>((const uint4*)v0, (uint4*)v1, nv)))
        COPYV("uint4 x4 nt, grid 2048", (copy16u_kernel<4, true><<<2048, 256, 0, st>>>((const uint4*)k0, (uint4*)k1, nk), copy16u_kernel<4, true><<<2048, 256, 0, st>>>((const uint4*)v0, (uint4*)v1, nv)))
        COPYV("uint4 x8 nt, grid 1024", (copy16u_kernel<8, true><<<1024, 256, 0, st>>>((const uint4*)k0, (uint4*)k1, nk), copy16u_kernel<8, true><<<1024, 256, 0, st>>>((const uint4*)v0, (uint4*)v1, nv)))
        COPYV("uint4 x4, one span per block", (copy16u_kernel<4, false><<<(unsigned)((nk + 1023) / 1024), 256, 0, st>>>((const uint4*)k0, (uint4*)k1, nk), copy16u_kernel<4, false><<<(unsigned)((nv + 1023) / 1024), 256, 0, st>>>((const uint4*)v0, (uint4*)v1, nv)))
        COPYV("tile shape (8 B + 4 B per lane)", (copy_tile_kernel<<<(n + 8191) / 8192, 512, 0, st>>>(k0, v0, k1, v1, n)))
    }
    {   // 8-byte records (u32 key + u32 value): what do 4-byte-per-lane accesses cost?
        const double gb8 = (double)n * 16.0 / 1e9;
#define COPY32(name, LD, ST) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float best = 1e9f; \
        for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0, st)); copy_tile32_kernel<LD, ST><<<n / 8192, 512, 0, st>>>((const u32*)k0, v0, (u32*)k1, v1, n); \
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
        printf("copy32 %-34s %8.3f ms  %7.1f GB/s\n", name, best, gb8 / best * 1e3); fflush(stdout); }
        COPY32("4 B loads, 4 B stores", false, false)
        COPY32("16 B loads, 4 B stores", true, false)
        COPY32("4 B loads, 16 B stores", false, true)
        COPY32("16 B loads, 16 B stores", true, true)
    }
    // calibration launches for the PMC counters (known byte counts at the kernel's access widths)
    copy8_kernel<<<4096, 256, 0, st>>>(k0, k1, (u64)n);
    copy4_kernel<<<4096, 256, 0, st>>>(v0, v1, (u64)n);
    CK(hipStreamSynchronize(st));
    const int shift = 24;   // a uniformly distributed digit
#define RUN(B, A, name) { float ms = run_pass<B, A>(ws, st, k0, v0, k1, v1, n, shift, 4); \
        printf("block %d abl %2d %-28s %8.3f ms  %7.1f GB/s\n", B, A, name, ms, gb / ms * 1e3); fflush(stdout); }
    RUN(256, 0, "full")
    RUN(256, 1, "no lookback")
    RUN(256, 16, "no next-hist")
    RUN(256, 17, "no lookback, no next-hist")
    RUN(256, 8, "linear stores")
    RUN(512, 0, "full")
    RUN(512, 1, "no lookback")
    RUN(512, 16, "no next-hist")
    RUN(512, 17, "no lookback, no next-hist")
    RUN(512, 8, "linear stores")
    RUN(512, 4, "no values")
#define RUNI(M, name) { float ms = run_pass<512, 0>(ws, st, k0, v0, k1, v1, n, shift, 4, 0, M); \
        printf("block 512 incl_mask %d %-24s %8.3f ms  %7.1f GB/s\n", M, name, ms, gb / ms * 1e3); fflush(stdout); }
    RUNI(0, "INCL every tile")
    RUNI(1, "INCL every 2nd tile")
    RUNI(3, "INCL every 4th tile")
    RUNI(7, "INCL every 8th tile")
    RUNI(0, "INCL every tile")
    RUN(512, 9, "linear stores, no lookback")
    RUN(512, 25, "linear, no lb, no nh")
    {   // correctness of the full pass: stable by digit, a permutation (values are iota), next-pass histogram
        std::vector<u64> ka(n); std::vector<u32> va(n), ha(NCHUNK * RADIX);
        CK(hipMemset(k1, 0, (size_t)n * 8)); CK(hipMemset(v1, 0xFF, (size_t)n * 4));
        run_pass<512, 0>(ws, st, k0, v0, k1, v1, n, shift, 1);
        CK(hipMemcpy(ka.data(), k1, (size_t)n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(va.data(), v1, (size_t)n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ha.data(), ws.hist(1), ha.size() * 4, hipMemcpyDeviceToHost));
        std::vector<u64> kin(n); CK(hipMemcpy(kin.data(), k0, (size_t)n * 8, hipMemcpyDeviceToHost));
        size_t unsorted = 0, wrongkey = 0;
        for (size_t i = 0; i < n; ++i) {
            wrongkey += va[i] >= n || kin[va[i]] != ka[i];
            if (i) { u32 da = (u32)(ka[i-1] >> shift) & 255u, db = (u32)(ka[i] >> shift) & 255u; unsorted += (da > db) || (da == db && va[i-1] >= va[i]); }
        }
        const SortGeom g = make_geom(n, 512 * SORT_ITEMS);
        std::vector<u32> href(NCHUNK * RADIX, 0);
        for (size_t i = 0; i < n; ++i) href[chunk_of_tile((u32)(i >> g.tile_shift), g.tpc) * RADIX + ((u32)(ka[i] >> (shift + 8)) & 255u)]++;
        size_t hbad = 0; for (size_t i = 0; i < href.size(); ++i) hbad += href[i] != ha[i];
        printf("check (512): %zu order violations, %zu records not matching their source, %zu next-histogram mismatches\n", unsorted, wrongkey, hbad);
    }
#define RUNH(B, A, H, name) { float ms = run_pass<B, A>(ws, st, k0, v0, k1, v1, n, shift, 4, H); \
        printf("block %d abl %2d home %d %-22s %8.3f ms  %7.1f GB/s\n", B, A, H, name, ms, gb / ms * 1e3); fflush(stdout); }
    RUNH(512, 17, 1, "no lb/nh, home=0")
    RUNH(512, 17, 2, "no lb/nh, home=bid&7")
    RUNH(512, 0, 1, "full, home=0")
    RUNH(512, 0, 2, "full, home=bid&7")
    RUNH(256, 17, 1, "no lb/nh, home=0")
    RUNH(256, 17, 2, "no lb/nh, home=bid&7")
    RUNH(256, 0, 1, "full, home=0")
    DeviceStatus ds; CK(hipMemcpy(&ds, ws.dstat, sizeof ds, hipMemcpyDeviceToHost));
    printf("device error flag: %u\n", ds.error);
    return 0;
}
