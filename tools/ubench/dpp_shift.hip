#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned lane = threadIdx.x;
    unsigned v = lane + 100;
    unsigned up1 = __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xF, 0xF, true);  // wave_shr:1 ?
    unsigned dn1 = __builtin_amdgcn_update_dpp(0u, v, 0x130, 0xF, 0xF, true);  // wave_shl:1 ?
    out[lane] = up1; out[64 + lane] = dn1;
}
int main() {
    unsigned *d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("up1:"); for (int i = 0; i < 64; i++) printf(" %u", h[i]); printf("\ndn1:"); for (int i = 0; i < 64; i++) printf(" %u", h[64+i]); printf("\n");
}
