// Micro-benchmark: does any cache-policy flavour of a random 16-byte load move less than a whole
// 128-byte line over the L2-miss path?  Same uniform dependent-chain pattern as gather3.hip, one
// chain per lane, load issued through inline asm with the given sc0/sc1/nt modifiers.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

#define DEFK(NAME, MODS, WIDTH, VT)                                                                   \
    __global__ void NAME(const uint8_t *table, uint32_t entries, int steps, uint32_t *out)               \
    {                                                                                                    \
        uint32_t idx = mix(blockIdx.x * blockDim.x + threadIdx.x + 12345u), acc = 0;                     \
        for (int s = 0; s < steps; s++) {                                                                \
            const uint8_t *p = table + (uint64_t)__umulhi(idx, entries) * 16;                            \
            VT v;                                                                                        \
            asm volatile("global_load_" WIDTH " %0, %1, off " MODS "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); \
            uint32_t x = *reinterpret_cast<uint32_t *>(&v);                                              \
            acc += x;                                                                                    \
            idx = mix(idx + x + 0x9e3779b9u);                                                            \
        }                                                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                \
    }

DEFK(k_plain, "", "dwordx4", u4)
DEFK(k_sc0, "sc0", "dwordx4", u4)
DEFK(k_sc1, "sc1", "dwordx4", u4)
DEFK(k_sc01, "sc0 sc1", "dwordx4", u4)
DEFK(k_nt, "nt", "dwordx4", u4)
DEFK(k_nt_sc0, "sc0 nt", "dwordx4", u4)
DEFK(k_nt_sc1, "sc1 nt", "dwordx4", u4)
DEFK(k_nt_sc01, "sc0 sc1 nt", "dwordx4", u4)
DEFK(k_dw, "", "dword", uint32_t)
DEFK(k_dw_nt, "nt", "dword", uint32_t)
DEFK(k_dw_sc01, "sc0 sc1", "dword", uint32_t)

typedef void (*kern_t)(const uint8_t *, uint32_t, int, uint32_t *);

void run(const char *name, kern_t k, const uint8_t *d_table, uint32_t entries, uint32_t *d_out)
{
    const int steps = 300, threads = 64, blocks = 256 * 32;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_table, entries, 30, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_table, entries, steps, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("table %8.1f MB  %-14s : %8.3f ms  %7.1f G lane-loads/s\n", entries * 16.0 / 1e6, name, ms,
           (double)blocks * threads * steps / ms / 1e6);
}

int main()
{
    uint32_t *d_out; CK(hipMalloc(&d_out, 256 * 32 * 64 * 4));
    for (size_t mb : {64, 1024}) {
        const uint32_t entries = (uint32_t)(mb * 1024 * 1024 / 16);
        std::vector<uint32_t> h((size_t)entries * 4);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
        uint8_t *d_table; CK(hipMalloc(&d_table, (size_t)entries * 16));
        CK(hipMemcpy(d_table, h.data(), (size_t)entries * 16, hipMemcpyHostToDevice));
        run("plain x4", k_plain, d_table, entries, d_out);
        run("sc0 x4", k_sc0, d_table, entries, d_out);
        run("sc1 x4", k_sc1, d_table, entries, d_out);
        run("sc0 sc1 x4", k_sc01, d_table, entries, d_out);
        run("nt x4", k_nt, d_table, entries, d_out);
        run("sc0 nt x4", k_nt_sc0, d_table, entries, d_out);
        run("sc1 nt x4", k_nt_sc1, d_table, entries, d_out);
        run("sc0 sc1 nt x4", k_nt_sc01, d_table, entries, d_out);
        run("plain dword", k_dw, d_table, entries, d_out);
        run("nt dword", k_dw_nt, d_table, entries, d_out);
        run("sc0 sc1 dword", k_dw_sc01, d_table, entries, d_out);
        CK(hipFree(d_table));
    }
    return 0;
}
