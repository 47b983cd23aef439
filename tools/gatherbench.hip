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

int main(int argc, char** argv) {
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
