// gatherbench.hip -- calibration of rocprofv3's FETCH_SIZE for RANDOM narrow reads (the access pattern of
// query_kernel): R lanes each read ONE 4-byte (or 8-byte) element at a pseudo-random index of a table much
// larger than the 256 MiB Infinity Cache, so every read is a distinct sector from HBM.  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- tools/gatherbench
// and divide the counter by R: the bytes the counter tallies per random read (profiles/r02_pmc_calibration.md).
//   hipcc -O3 --offload-arch=gfx950 -o tools/gatherbench tools/gatherbench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T* __restrict__ table, uint64_t elems, uint64_t reads, uint64_t salt,
                                                     T* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= reads) return;
    out[i] = table[mix(i ^ salt) % elems];
}
__global__ void fill_kernel(uint32_t* p, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (uint32_t)i;
}

// ---- sweep (round 3): the random-request ceiling as a CURVE, not a point -------------------------------------------
// `lanes` lanes are resident (grid = lanes / 256, every lane loops over its share of `reads` independent requests, four in
// flight per lane); each request reads BYTES (4, 64 or 128) from a random BYTES-aligned place of a table of `footprint`
// bytes.  tools/gatherbench sweep  prints requests/s and the sector bytes/s they amount to for every combination.
template <int BYTES>
__global__ __launch_bounds__(256) void gather_sweep_kernel(const uint4* __restrict__ table, uint64_t units, uint64_t per_lane, uint64_t salt,
                                                           uint32_t* __restrict__ out) {
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    const uint64_t base = lane * per_lane;
    for (uint64_t r = 0; r < per_lane; r += 4) {
        uint4 v[4][BYTES >= 16 ? BYTES / 16 : 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t u = mix((base + r + k) ^ salt) % units;
            if (BYTES == 4) { v[k][0].x = reinterpret_cast<const uint32_t*>(table)[u]; v[k][0].y = v[k][0].z = v[k][0].w = 0; }
            else {
#pragma unroll
                for (int q = 0; q < BYTES / 16; ++q) v[k][q] = table[u * (BYTES / 16) + q];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < (BYTES >= 16 ? BYTES / 16 : 1); ++q) acc ^= v[k][q].x ^ v[k][q].y ^ v[k][q].z ^ v[k][q].w;
    }
    out[lane] = acc;
}

static int sweep() {
    const uint64_t max_bytes = 16ull << 30;
    void* table; void* out;
    CK(hipMalloc(&table, max_bytes));
    CK(hipMalloc(&out, (1ull << 24) * 4));
    fill_kernel<<<4096, 256>>>((uint32_t*)table, max_bytes / 4);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint64_t reads = 1ull << 26;
    printf("footprint_MiB request_B resident_lanes  ms  G_requests_per_s  sector_GB_per_s\n");
    for (uint64_t fp : {64ull << 20, 256ull << 20, 1ull << 30, 4ull << 30, 16ull << 30}) {
        for (int bytes : {4, 64, 128}) {
            for (uint64_t lanes : {1ull << 16, 1ull << 18, 1ull << 20, 1ull << 22, 1ull << 24}) {
                const uint64_t per_lane = reads / lanes;
                const uint64_t units = fp / (uint64_t)bytes;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(a));
                    const unsigned grid = (unsigned)(lanes / 256);
                    if (bytes == 4) gather_sweep_kernel<4><<<grid, 256>>>((const uint4*)table, units, per_lane, 99 + rep, (uint32_t*)out);
                    else if (bytes == 64) gather_sweep_kernel<64><<<grid, 256>>>((const uint4*)table, units, per_lane, 99 + rep, (uint32_t*)out);
                    else gather_sweep_kernel<128><<<grid, 256>>>((const uint4*)table, units, per_lane, 99 + rep, (uint32_t*)out);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                const double rps = reads / (best * 1e-3);
                printf("%8llu %6d %10llu %8.3f %8.2f %9.1f\n", (unsigned long long)(fp >> 20), bytes, (unsigned long long)lanes, best, rps / 1e9,
                       rps * (bytes < 64 ? 64 : bytes) / 1e9);
            }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "sweep")) return sweep();
    const uint64_t bytes = 4ull << 30;
    const uint64_t reads = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 24);
    void* table; void* out;
    CK(hipMalloc(&table, bytes));
    CK(hipMalloc(&out, reads * 8));
    fill_kernel<<<2048, 256>>>((uint32_t*)table, bytes / 4);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        float ms4 = 0, ms8 = 0;
        CK(hipEventRecord(a));
        gather_kernel<uint32_t><<<(unsigned)((reads + 255) / 256), 256>>>((const uint32_t*)table, bytes / 4, reads, 1234567ull * (rep + 1), (uint32_t*)out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms4, a, b));
        CK(hipEventRecord(a));
        gather_kernel<uint64_t><<<(unsigned)((reads + 255) / 256), 256>>>((const uint64_t*)table, bytes / 8, reads, 7654321ull * (rep + 1), (uint64_t*)out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms8, a, b));
        printf("reads %llu  4-byte: %.3f ms (%.2f G reads/s)  8-byte: %.3f ms (%.2f G reads/s)\n", (unsigned long long)reads, ms4,
               reads / ms4 / 1e6, ms8, reads / ms8 / 1e6);
    }
    return 0;
}
