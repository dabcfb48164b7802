// tools/mfma_probe.hip -- microbenchmark: v_mfma_f32_32x32x2_f32 issue patterns on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_probe tools/mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0)
{
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS> void run(int wg_per_cu, int iters)
{
    int cus = 256;
    float* out; hipMalloc(&out, sizeof(float) * cus * wg_per_cu * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<CHAINS>, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)cus * wg_per_cu * 4 /*waves*/ * iters * 8.0 * CHAINS * 4096.0;
        if (rep == 2) printf("chains=%d wg/cu=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", CHAINS, wg_per_cu, iters, ms, flop / ms / 1e9);
    }
    hipFree(out);
}

int main()
{
    run<1>(1, 20000); run<1>(2, 10000); run<1>(4, 5000);
    run<2>(1, 10000); run<2>(2, 5000);
    run<4>(1, 5000); run<4>(2, 2500);
    return 0;
}
