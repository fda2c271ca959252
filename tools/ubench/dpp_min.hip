// wave-wide minimum by data-parallel primitives (no LDS crossbar): checked against the plain minimum on random data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned wave_min_dpp(unsigned v)
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false)); // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false)); // row_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false)); // row_bcast:15 into rows 1, 3
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false)); // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__global__ void k(const unsigned *in, unsigned *out) { out[blockIdx.x * 64 + threadIdx.x] = wave_min_dpp(in[blockIdx.x * 64 + threadIdx.x]); }
int main()
{
    const int W = 4096;
    unsigned *h = (unsigned *)malloc(W * 64 * 4), *r = (unsigned *)malloc(W * 64 * 4), *di, *dout;
    for (int i = 0; i < W * 64; i++) h[i] = (unsigned)rand();
    hipMalloc(&di, W * 64 * 4); hipMalloc(&dout, W * 64 * 4);
    hipMemcpy(di, h, W * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, di, dout);
    hipMemcpy(r, dout, W * 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < W; w++) {
        unsigned m = ~0u;
        for (int l = 0; l < 64; l++) m = h[w * 64 + l] < m ? h[w * 64 + l] : m;
        for (int l = 0; l < 64; l++) bad += r[w * 64 + l] != m;
    }
    printf("wave_min_dpp: %d wrong of %d\n", bad, W * 64);
    return bad != 0;
}
