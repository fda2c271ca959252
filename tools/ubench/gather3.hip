// Micro-benchmark: uniformly random dependent 16-byte gathers (the walk kernel's rank-block loads when
// the index does not fit L2).  Each lane runs CH independent dependent chains; the next address is a
// murmur-style hash of the loaded value, so the pattern is uniform over the table (gather.hip's LCG is not).
// Reports lane-loads/s for table sizes from L2-sized to HBM-sized.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

template <int CH>
__global__ void gather_kernel(const uint4 *table, uint32_t entries, int steps, uint32_t *out)
{
    uint32_t idx[CH], acc = 0;
    for (int c = 0; c < CH; c++) idx[c] = mix((blockIdx.x * blockDim.x + threadIdx.x) * CH + c + 12345u);
    for (int s = 0; s < steps; s++) {
        uint4 v[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = table[__umulhi(idx[c], entries)];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            acc += v[c].y;
            idx[c] = mix(idx[c] + v[c].x + 0x9e3779b9u);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int CH>
void run(const uint4 *d_table, uint32_t entries, int waves_per_cu, uint32_t *d_out)
{
    const int cus = 256, steps = 400, threads = 64, blocks = cus * waves_per_cu;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<CH>), dim3(blocks), dim3(threads), 0, 0, d_table, entries, 50, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<CH>), dim3(blocks), dim3(threads), 0, 0, d_table, entries, steps, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double loads = (double)blocks * threads * steps * CH;
    printf("table %9.1f MB  waves/CU %2d  chains/lane %d : %8.3f ms  %7.1f G lane-loads/s  (x128 B = %5.2f TB/s, x64 B = %5.2f TB/s)  %6.0f ns/step\n",
           entries * 16.0 / 1e6, waves_per_cu, CH, ms, loads / ms / 1e6, loads * 128 / ms / 1e9, loads * 64 / ms / 1e9,
           ms * 1e6 / steps);
}

int main(int argc, char **argv)
{
    uint32_t *d_out; CK(hipMalloc(&d_out, 256 * 32 * 64 * 4));
    const size_t max_mb = argc > 1 ? atol(argv[1]) : 4096;
    for (size_t mb : {2, 16, 64, 128, 512, 2048, 4096}) {
        if (mb > max_mb) break;
        const uint32_t entries = (uint32_t)(mb * 1024 * 1024 / 16);
        std::vector<uint32_t> h((size_t)entries * 4);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
        uint4 *d_table; CK(hipMalloc(&d_table, (size_t)entries * 16));
        CK(hipMemcpy(d_table, h.data(), (size_t)entries * 16, hipMemcpyHostToDevice));
        for (int w : {16, 32}) {
            run<1>(d_table, entries, w, d_out);
            run<2>(d_table, entries, w, d_out);
            run<4>(d_table, entries, w, d_out);
        }
        CK(hipFree(d_table));
    }
    return 0;
}
