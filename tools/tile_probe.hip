// tools/tile_probe.hip -- the tile functions of hscmp_mfma.h alone: 16 waves per CU (4 per SIMD) run 32-position tiles back to
// back against the 64 KB dictionary image in LDS, nothing else.  Reports the fraction of the fp32 matrix peak each form
// of the tile reaches by itself (what the greedy loop's tile phase can reach at best).
// Measured (r02): lean tile 94.9 % of the fp32 matrix peak at 4 and at 2 waves per SIMD, full tile 93.2 % at 2; a variant of
// the lean tile with the 32 B operands in registers: 95.1 %.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I hierarchical-sparse-coding_amd/csrc -I include -o tools/tile_probe tools/tile_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "hscmp_device.h"
#include "hscmp_kernels.h"
#include "hscmp_mfma.h"
using namespace hscmp;

template <int FORM, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD, WAVES_PER_SIMD) void tiles(const float* img, float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dimg = reinterpret_cast<float*>(smem);
    float* win = dimg + 8 * 8 * 256;                      // G = 8 groups, S4 = 8 chunks, 256 floats each = 64 KB
    for (int i = threadIdx.x; i < 8 * 8 * 256; i += blockDim.x) dimg[i] = img[i];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) win[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f; int gacc = 0;
    for (int it = 0; it < iters; ++it) {
        float sc;
        int grp;
        const float* w = win + 224 * (wave & 15) + 32 * (it & 3);
        if constexpr (FORM == 0) sc = mfma_tile_score_lean<8, false>(dimg, w, nullptr, 8, lane, grp);
        else sc = mfma_tile_score<8, false>(dimg, w, nullptr, 8, 8, lane, grp);
        acc += sc; gacc += grp;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)gacc;
}

template <int FORM, int WPS> void run(const char* name, const float* img, float* out)
{
    const int iters = 2000;
    auto kern = tiles<FORM, WPS>;
    const size_t lds = 65536 + 4096 * 4;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256 * WPS), lds, 0, img, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 256.0 * (4 * WPS) * iters * 256.0 * 4096.0;
    printf("%-46s %d waves/SIMD: %8.3f ms  %6.1f TFLOP/s  %.1f%% of 157.3\n", name, WPS, best, flop / best / 1e9, flop / best / 1e9 / 157.3 * 100);
}

int main()
{
    float* img; float* out;
    hipMalloc(&img, 65536); hipMemset(img, 0, 65536);
    hipMalloc(&out, 256 * 1024 * 4);
    run<0, 4>("lean tile (B in registers, A one chunk ahead)", img, out);
    run<0, 3>("lean tile (B in registers, A one chunk ahead)", img, out);
    run<0, 1>("lean tile (B in registers, A one chunk ahead)", img, out);
    run<1, 2>("full tile (B in registers, A a group ahead)", img, out);
    run<0, 2>("lean tile (B in registers, A one chunk ahead)", img, out);
    return 0;
}
