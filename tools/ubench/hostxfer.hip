// Micro-benchmark behind the host-buffer path: what does it cost to get 600 MB of
// pageable user memory to the device and back?  (a) hipHostRegister in place, (b) hipMemcpy straight
// from pageable memory, (c) threaded memcpy into a pinned staging buffer + DMA.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fflush(stdout); printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void copy_kernel(const uint4 *src, uint4 *dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void spin_kernel(uint32_t *out, long cycles, const uint4 *table, uint32_t mask)
{
    const long t0 = clock64();
    uint32_t acc = 0, idx = threadIdx.x + blockIdx.x * 64;
    while (clock64() - t0 < cycles) { // dependent random 16-byte loads, like the walk kernel
        const uint4 v = table[idx & mask];
        acc += v.x;
        idx = idx * 1664525u + 1013904223u + v.y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static void par_memcpy(void *dst, const void *src, size_t n, int threads)
{
    std::vector<std::thread> t;
    const size_t per = (n / threads + 4095) & ~size_t(4095);
    for (int i = 0; i < threads; i++) {
        const size_t a = std::min(n, per * i), b = std::min(n, per * (i + 1));
        if (a < b) t.emplace_back([=] { std::memcpy((char *)dst + a, (const char *)src + a, b - a); });
    }
    for (auto &x : t) x.join();
}

int main()
{
    const size_t N = 600ull << 20;
    uint8_t *user = (uint8_t *)malloc(N), *user2 = (uint8_t *)malloc(N);
    memset(user, 1, N); memset(user2, 2, N);
    void *d; CK(hipMalloc(&d, N));
    void *pin; CK(hipHostMalloc(&pin, N, hipHostMallocDefault));
    memset(pin, 3, N);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipHostRegister(user, N, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(d, user, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipMemcpyAsync(user, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        double t3 = now();
        CK(hipHostUnregister(user));
        double t4 = now();
        fflush(stdout); printf("register %6.1f ms | H2D registered %6.1f ms (%5.1f GB/s) | D2H registered %6.1f ms (%5.1f GB/s) | unregister %6.1f ms\n",
               (t1 - t0) * 1e3, (t2 - t1) * 1e3, N / (t2 - t1) / 1e9, (t3 - t2) * 1e3, N / (t3 - t2) / 1e9, (t4 - t3) * 1e3);
        t0 = now();
        CK(hipMemcpy(d, user2, N, hipMemcpyHostToDevice));
        t1 = now();
        CK(hipMemcpy(user2, d, N, hipMemcpyDeviceToHost));
        t2 = now();
        fflush(stdout); printf("pageable H2D %6.1f ms (%5.1f GB/s) | pageable D2H %6.1f ms (%5.1f GB/s)\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9,
               (t2 - t1) * 1e3, N / (t2 - t1) / 1e9);
        t0 = now();
        CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        t1 = now();
        CK(hipMemcpyAsync(pin, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        t2 = now();
        fflush(stdout); printf("pinned H2D %6.1f ms (%5.1f GB/s) | pinned D2H %6.1f ms (%5.1f GB/s)\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9,
               (t2 - t1) * 1e3, N / (t2 - t1) / 1e9);
        for (int th : {1, 4, 8, 16, 32}) {
            t0 = now();
            par_memcpy(pin, user2, N, th);
            t1 = now();
            par_memcpy(user2, pin, N, th);
            t2 = now();
            fflush(stdout); printf("  memcpy %2d threads: user->pinned %6.1f ms (%5.1f GB/s) | pinned->user %6.1f ms (%5.1f GB/s)\n", th,
                   (t1 - t0) * 1e3, N / (t1 - t0) / 1e9, (t2 - t1) * 1e3, N / (t2 - t1) / 1e9);
        }
    }
    // both directions at once (two streams), pinned
    hipStream_t s2; CK(hipStreamCreate(&s2));
    void *d2; CK(hipMalloc(&d2, N));
    void *pin2; CK(hipHostMalloc(&pin2, N, hipHostMallocDefault));
    double t0 = now();
    CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(pin2, d2, N, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
    double t1 = now();
    fflush(stdout); printf("pinned H2D + D2H concurrently: %6.1f ms (%5.1f GB/s each way)\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
    // does an H2D copy make progress while a kernel holds every wave slot?
    {
        uint32_t *d_out; CK(hipMalloc(&d_out, 256 * 32 * 64 * 4));
        hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        for (int wpc : {32, 24, 8}) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            hipLaunchKernelGGL(spin_kernel, dim3(256 * wpc), dim3(64), 0, s2, d_out, 20000000L, (const uint4 *)d2, (uint32_t)(N / 16 / 2 - 1));
            CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            double t1 = now();
            CK(hipStreamSynchronize(s2));
            double t2 = now();
            fflush(stdout); printf("spin kernel %2d waves/CU: H2D copy done after %6.1f ms, kernel done after %6.1f ms\n", wpc, (t1 - t0) * 1e3, (t2 - t0) * 1e3);
        }
    }
    // D2H by a kernel storing into pinned host memory, alone and next to an SDMA H2D copy
    for (int blocks : {256, 1024, 4096}) {
        t0 = now();
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d2, (uint4 *)pin2, N / 16);
        CK(hipStreamSynchronize(s2));
        t1 = now();
        fflush(stdout); printf("kernel D2H (%4d blocks) alone: %6.1f ms (%5.1f GB/s)\n", blocks, (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        t0 = now();
        CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d2, (uint4 *)pin2, N / 16);
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
        t1 = now();
        fflush(stdout); printf("kernel D2H (%4d blocks) + SDMA H2D concurrently: %6.1f ms (%5.1f GB/s each way)\n", blocks, (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        t0 = now();
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, (const uint4 *)pin, (uint4 *)d, N / 16);
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d2, (uint4 *)pin2, N / 16);
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
        t1 = now();
        fflush(stdout); printf("kernel D2H + kernel H2D (%4d blocks each) concurrently: %6.1f ms (%5.1f GB/s each way)\n", blocks, (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
    }
    return 0;
}
