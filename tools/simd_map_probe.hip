// which SIMD does wave w of a 1024-thread workgroup run on? (HW_ID register, gfx9 layout: simd_id = bits 5:4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned* d; hipMalloc(&d, 4 * 16 * 512);
    hipLaunchKernelGGL(k, dim3(512), dim3(1024), 0, 0, d);
    unsigned h[16 * 512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[16][4] = {{0}};
    for (int b = 0; b < 512; ++b) for (int w = 0; w < 16; ++w) hist[w][(h[b * 16 + w] >> 4) & 3]++;
    for (int w = 0; w < 16; ++w) printf("wave %2d: SIMD histogram over 512 workgroups: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 3; ++b) { printf("wg %d:", b); for (int w = 0; w < 16; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3); printf("\n"); }
    return 0;
}
