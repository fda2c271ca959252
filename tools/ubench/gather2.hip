// Does a partially-masked VMEM instruction cost as much as a full one?  (walk kernel question)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: 2x16B gathers only.  1: + dword load by 1/4 of the lanes each step.  2: + dword store by 1/4 lanes.
// 3: both.  4: both but by ALL lanes every 4th step (same bytes, fewer instructions). 5: extra VALU filler (100 ops)
template <int MODE>
__global__ void k(const uint4 *table, uint32_t mask, int steps, const uint32_t *stream, uint32_t *sink, uint32_t *out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = tid * 2654435761u, acc = 0;
    const uint32_t *my_in = stream + (size_t)tid * 64;
    uint32_t *my_out = sink + (size_t)tid * 64;
    for (int s = 0; s < steps; s++) {
        uint32_t i0 = idx & mask;
        uint4 a = table[i0];
        uint4 b = table[(i0 + 1) & mask];
        uint32_t v = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.w;
        if (MODE == 1 || MODE == 3) { if (((tid + s) & 3) == 0) acc += my_in[(s >> 2) & 63]; }
        if (MODE == 2 || MODE == 3) { if (((tid + s) & 3) == 1) my_out[(s >> 2) & 63] = acc; }
        if (MODE == 4) { if ((s & 3) == 0) { acc += my_in[(s >> 2) & 63]; my_out[(s >> 2) & 63] = acc; } }
        if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 50; j++) { v = v * 3 + (v >> 7); v ^= acc + j; }
        }
        acc += v;
        idx = idx * 1664525u + 1013904223u + v;
    }
    out[tid] = acc;
}

template <int MODE> void run(const char *name, const uint4 *t, uint32_t entries, const uint32_t *stream, uint32_t *sink, uint32_t *out)
{
    const int blocks = 256 * 32, threads = 64, steps = 1000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, t, entries - 1, 50, stream, sink, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, t, entries - 1, steps, stream, sink, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s table %6.2f MB: %7.3f ms  %6.1f G lane-steps/s  (%.0f cyc/wave-step/CU)\n", name, entries * 16.0 / 1e6, ms,
           (double)blocks * threads * steps / ms / 1e6, ms * 1e-3 * 2.3e9 / (32.0 * steps));
}

int main()
{
    const size_t lanes = 256 * 32 * 64;
    uint32_t *out, *stream, *sink;
    CK(hipMalloc(&out, lanes * 4)); CK(hipMalloc(&stream, lanes * 64 * 4)); CK(hipMalloc(&sink, lanes * 64 * 4));
    CK(hipMemset(stream, 1, lanes * 64 * 4));
    for (uint32_t log2e : {11u, 18u}) {
        uint32_t entries = 1u << log2e;
        std::vector<uint32_t> h(entries * 4);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) >> 7;
        uint4 *t; CK(hipMalloc(&t, entries * 16));
        CK(hipMemcpy(t, h.data(), entries * 16, hipMemcpyHostToDevice));
        run<0>("2x16B gathers", t, entries, stream, sink, out);
        run<1>("+ dword load, 1/4 lanes per step", t, entries, stream, sink, out);
        run<2>("+ dword store, 1/4 lanes per step", t, entries, stream, sink, out);
        run<3>("+ both, 1/4 lanes per step", t, entries, stream, sink, out);
        run<4>("+ both, all lanes every 4th step", t, entries, stream, sink, out);
        run<5>("2x16B gathers + 100 VALU", t, entries, stream, sink, out);
        CK(hipFree(t));
    }
    return 0;
}
