// Micro-benchmark: gather3.hip's uniformly random dependent 16-byte gathers on tables of 1 ... 48 GB (the 1 Gbp / 3 Gbp
// indexes of DESIGN.md section 6): does the L2-miss path still deliver its ~56 G fills/s when the table spans tens of
// GB (address translation)?  The table is filled on the device.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

__global__ void fill_kernel(uint4 *table, uint64_t entries)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < entries; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t h = mix((uint32_t)i * 2654435761u + (uint32_t)(i >> 32));
        table[i] = make_uint4(h, mix(h), mix(h + 1u), mix(h + 2u));
    }
}

__global__ void gather_kernel(const uint4 *table, uint32_t entries, int steps, uint32_t *out)
{
    uint32_t idx = mix(blockIdx.x * blockDim.x + threadIdx.x + 12345u), acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4 v = table[__umulhi(idx, entries)];
        acc += v.y;
        idx = mix(idx + v.x + 0x9e3779b9u);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    uint32_t *d_out; CK(hipMalloc(&d_out, 256 * 32 * 64 * 4));
    const size_t max_gb = argc > 1 ? atol(argv[1]) : 48;
    for (size_t gb : {1, 4, 16, 32, 48}) {
        if (gb > max_gb) break;
        const uint64_t entries64 = gb * 1024ull * 1024 * 1024 / 16;
        const uint32_t entries = (uint32_t)entries64;
        uint4 *d_table; CK(hipMalloc(&d_table, entries64 * 16));
        hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, d_table, entries64);
        CK(hipDeviceSynchronize());
        for (int w : {8, 16, 32}) {
            const int steps = 300, threads = 64, blocks = 256 * w;
            hipEvent_t a, b;
            CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(threads), 0, 0, d_table, entries, 40, d_out);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(threads), 0, 0, d_table, entries, steps, d_out);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double loads = (double)blocks * threads * steps;
            printf("table %5zu GB  waves/CU %2d : %8.3f ms  %6.1f G lane-loads/s  %6.0f ns/step\n", gb, w, ms, loads / ms / 1e6, ms * 1e6 / steps);
        }
        CK(hipFree(d_table));
    }
    return 0;
}
