// tools/mfma_f64_probe.hip -- is v_mfma_f64_16x16x4_f64 a k-ordered fma chain (bit-exact)? what rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// C[16x16] = sum over 8 k-steps of 4: A[i][k] (16 x 32), B[k][j] (32 x 16); one wave
__global__ void exact_probe(const double* A, const double* B, double* C)
{
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    f64x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < 8; ++s) {
        const double a = A[i * 32 + 4 * s + kk];          // A[i][k], k = 4s + kk
        const double b = B[(4 * s + kk) * 16 + i];        // B[k][j], j = lane & 15
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];    // row = (lane>>4) + 4r, col = lane&15
}

template <int CHAINS>
__global__ __launch_bounds__(256) void rate_probe(double* out, int iters)
{
    f64x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f64x4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS> void run_rate(int wg_per_cu, int iters)
{
    double* out; hipMalloc(&out, sizeof(double) * 256 * wg_per_cu * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_probe<CHAINS>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 256.0 * wg_per_cu * 4 * iters * 8.0 * CHAINS * 2048.0;
    printf("f64 16x16x4: chains=%d wg/cu=%d: %.3f ms  %.1f TFLOP/s\n", CHAINS, wg_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}

int main()
{
    std::mt19937_64 rng(7); std::normal_distribution<double> nd(0, 1);
    std::vector<double> A(16 * 32), B(32 * 16), C(256), R(256);
    for (auto& v : A) v = nd(rng) * std::exp(nd(rng) * 3);
    for (auto& v : B) v = nd(rng) * std::exp(nd(rng) * 3);
    double *dA, *dB, *dC; hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 256 * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0; double maxrel = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double acc = 0; for (int k = 0; k < 32; ++k) acc = fma(A[i * 32 + k], B[k * 16 + j], acc);
        if (acc != C[i * 16 + j]) { ++bad; maxrel = fmax(maxrel, fabs(acc - C[i * 16 + j]) / fabs(acc)); }
    }
    printf("f64 MFMA vs sequential fma chain: %d of 256 differ (max rel %.3g)\n", bad, maxrel);
    run_rate<1>(1, 4000); run_rate<1>(2, 2000); run_rate<2>(1, 2000); run_rate<4>(1, 1000); run_rate<4>(2, 500);
    return 0;
}
