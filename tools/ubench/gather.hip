// Micro-benchmark: rate of divergent per-lane gathers (what the walk kernel's rank loads are).
// Each lane runs a dependent chain of random loads from a table; reports lane-loads/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BYTES, int PAIR>   // BYTES per lane per load (4 or 16); PAIR: loads per step (1 or 2)
__global__ void gather_kernel(const uint4 *table, uint32_t mask, int steps, uint32_t *out)
{
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        uint32_t i0 = idx & mask;
        uint32_t v;
        if (BYTES == 16) {
            uint4 a = table[i0];
            v = a.x ^ a.y ^ a.z ^ a.w;
            if (PAIR == 2) { uint4 b = table[(i0 + 1) & mask]; v ^= b.x ^ b.w; }        // neighbouring block (same/adjacent line)
            if (PAIR == 3) { uint4 b = table[(i0 ^ 0x55555) & mask]; v ^= b.x ^ b.w; }  // unrelated second line
        } else {
            v = reinterpret_cast<const uint32_t *>(table)[i0 * 4];
            if (PAIR == 2) v ^= reinterpret_cast<const uint32_t *>(table)[((i0 + 1) & mask) * 4];
            if (PAIR == 3) v ^= reinterpret_cast<const uint32_t *>(table)[((i0 ^ 0x55555) & mask) * 4];
        }
        acc += v;
        idx = idx * 1664525u + 1013904223u + v;   // dependent chain
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int BYTES, int PAIR>
void run(const char *name, const uint4 *d_table, uint32_t entries, int waves_per_cu, uint32_t *d_out)
{
    const int cus = 256, steps = 2000;
    const int threads = 64, blocks = cus * waves_per_cu;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<BYTES, PAIR>), dim3(blocks), dim3(threads), 0, 0, d_table, entries - 1, 100, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<BYTES, PAIR>), dim3(blocks), dim3(threads), 0, 0, d_table, entries - 1, steps, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double lane_steps = (double)blocks * threads * steps;
    int loads = PAIR == 1 ? 1 : 2;
    printf("%-26s table %8.2f MB  waves/CU %2d : %7.3f ms  %7.1f G lane-steps/s  %7.1f G lane-loads/s  (%.0f cyc/wave-step/CU @2.3GHz)\n",
           name, entries * 16.0 / 1e6, waves_per_cu, ms, lane_steps / ms / 1e6, lane_steps * loads / ms / 1e6,
           ms * 1e-3 * 2.3e9 / ((double)waves_per_cu * steps));
}

int main()
{
    uint32_t *d_out; CK(hipMalloc(&d_out, 256 * 32 * 64 * 4));
    for (uint32_t log2e : {11u, 16u, 18u, 21u, 24u}) {   // 32 KB, 1 MB, 4 MB, 32 MB, 256 MB
        uint32_t entries = 1u << log2e;
        std::vector<uint32_t> h(entries * 4);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) >> 7;
        uint4 *d_table; CK(hipMalloc(&d_table, entries * 16));
        CK(hipMemcpy(d_table, h.data(), entries * 16, hipMemcpyHostToDevice));
        for (int w : {8, 32}) {
            run<16, 1>("16B x1", d_table, entries, w, d_out);
            run<16, 2>("16B x2 (adjacent)", d_table, entries, w, d_out);
            run<16, 3>("16B x2 (independent)", d_table, entries, w, d_out);
            run<4, 1>("4B x1", d_table, entries, w, d_out);
            run<4, 3>("4B x2 (independent)", d_table, entries, w, d_out);
        }
        CK(hipFree(d_table));
    }
    return 0;
}
