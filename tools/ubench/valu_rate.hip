// Micro-benchmark: issue cost of the integer VALU instructions the walk kernel leans on.
// 8 waves per SIMD, 4 independent chains per lane, 64 instructions of one kind per loop trip.
// Prints cycles per wave-instruction per SIMD (4 = full rate on a 16-lane SIMD).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)

#define DEFK(NAME, ASM)                                                                       \
    __global__ void NAME(uint32_t *out, int trips, uint32_t s)                                \
    {                                                                                         \
        uint32_t a = threadIdx.x + 1, b = a * 3, c = a * 5, d = a * 7;                        \
        uint64_t e = a, f = b;                                                                \
        for (int t = 0; t < trips; t++) {                                                     \
            REP16(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(s) : "s10", "s11", "s12", "s13", "vcc");) \
        }                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (uint32_t)e + (uint32_t)f; \
    }

// each ASM string = 4 instructions on 4 independent registers
DEFK(k_add, "v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6")
DEFK(k_mul_lo, "v_mul_lo_u32 %0, %0, %6\n v_mul_lo_u32 %1, %1, %6\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %6")
DEFK(k_mul_hi, "v_mul_hi_u32 %0, %0, %6\n v_mul_hi_u32 %1, %1, %6\n v_mul_hi_u32 %2, %2, %6\n v_mul_hi_u32 %3, %3, %6")
DEFK(k_mul24, "v_mul_u32_u24 %0, %0, %6\n v_mul_u32_u24 %1, %1, %6\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %6")
DEFK(k_mad24, "v_mad_u32_u24 %0, %0, %6, %1\n v_mad_u32_u24 %1, %1, %6, %2\n v_mad_u32_u24 %2, %2, %6, %3\n v_mad_u32_u24 %3, %3, %6, %0")
DEFK(k_mad64, "v_mad_u64_u32 %4, vcc, %0, %6, %4\n v_mad_u64_u32 %5, vcc, %1, %6, %5\n v_mad_u64_u32 %4, vcc, %2, %6, %4\n v_mad_u64_u32 %5, vcc, %3, %6, %5")
DEFK(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %1, %1, %2\n v_bcnt_u32_b32 %2, %2, %3\n v_bcnt_u32_b32 %3, %3, %0")
DEFK(k_lshl64, "v_lshlrev_b64 %4, %0, %4\n v_lshlrev_b64 %5, %1, %5\n v_lshlrev_b64 %4, %2, %4\n v_lshlrev_b64 %5, %3, %5")
DEFK(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc")
DEFK(k_cmp_s, "v_cmp_lt_u32 s[10:11], %0, %1\n v_cmp_lt_u32 s[12:13], %1, %2\n v_cmp_lt_u32 s[10:11], %2, %3\n v_cmp_lt_u32 s[12:13], %3, %0")
DEFK(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %1, %1, 3, %2\n v_lshl_add_u32 %2, %2, 3, %3\n v_lshl_add_u32 %3, %3, 3, %0")
DEFK(k_add3, "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1")
DEFK(k_ffbh, "v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3")

typedef void (*kern_t)(uint32_t *, int, uint32_t);

void run(const char *name, kern_t k, uint32_t *d_out, int waves_per_simd)
{
    const int trips = 2000, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, 10, 3u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, trips, 3u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_simd = (double)waves_per_simd * trips * 64;
    printf("%-10s waves/SIMD %d : %7.3f ms  %5.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
    fflush(stdout);
}

int main()
{
    uint32_t *d_out; CK(hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4));
    for (int w : {8, 4}) {
        run("add", k_add, d_out, w); run("mul_lo", k_mul_lo, d_out, w); run("mul_hi", k_mul_hi, d_out, w);
        run("mul24", k_mul24, d_out, w); run("mad24", k_mad24, d_out, w); run("mad_u64", k_mad64, d_out, w);
        run("bcnt", k_bcnt, d_out, w); run("lshl_b64", k_lshl64, d_out, w); run("cndmask", k_cndmask, d_out, w);
        run("cmp->sgpr", k_cmp_s, d_out, w); run("lshl_add", k_lshl_add, d_out, w); run("add3", k_add3, d_out, w);
        run("ffbh", k_ffbh, d_out, w);
    }
    return 0;
}
