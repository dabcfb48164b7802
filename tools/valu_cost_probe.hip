// tools/valu_cost_probe.hip -- what does one non-MFMA instruction cost a SIMD that is otherwise saturated with
// v_mfma_f32_32x32x2_f32 (64 cycles each)?  16 waves per CU (4 per SIMD, as the four-signal greedy loop runs).
//   mode 0: every wave runs blocks of 32 MFMAs followed by NV instructions of one KIND
//   mode 1: half of the waves run MFMAs only, the other half the NV instructions only (cross-wave interference)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/valu_cost_probe tools/valu_cost_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND> __device__ __forceinline__ void op(float& x, float& y, int& i, int& j, unsigned lds)
{
    if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if constexpr (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i) : "v"(j));
    if constexpr (KIND == 2) asm volatile("v_max_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
    if constexpr (KIND == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(j) : "v"(lds));
    if constexpr (KIND == 4) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
    if constexpr (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if constexpr (KIND == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(i) : "v"(j));
    if constexpr (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(i) : "v"(j));
    if constexpr (KIND == 8) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(i) : "v"(j));
    if constexpr (KIND == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");
    if constexpr (KIND == 10) asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(i) : "s20");
    if constexpr (KIND == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&x) : "v"(1.0));   // (x,y adjacent not guaranteed: own regs below)
}

template <int KIND, int NV, int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, float a0, float b0)
{
    __shared__ float buf[1024];
    buf[threadIdx.x] = a0;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
    float x = a, y = b; int i = threadIdx.x, j = 3;
    const unsigned lds = (unsigned)(threadIdx.x * 4);
    const bool mf = MODE == 0 || ((threadIdx.x >> 6) & 4) == 0;      // mode 1: waves 0-3, 8-11 MFMA; 4-7, 12-15 the other kind
    const bool va = MODE == 0 || !mf;
    for (int it = 0; it < iters; ++it) {
        if (mf) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        }
        if (va) {
#pragma unroll
            for (int u = 0; u < NV; ++u) op<KIND>(x, y, i, j, lds);
            if constexpr (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float s = x + (float)i + (float)j;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out;
template <int KIND, int NV, int MODE> float run(int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KIND, NV, MODE>), dim3(256), dim3(1024), 0, 0, g_out, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int KIND, int MODE> void sweep(const char* name, int iters, double ghz)
{
    const float t0 = run<KIND, 0, MODE>(iters), t8 = run<KIND, 8, MODE>(iters), t32 = run<KIND, 32, MODE>(iters), t64 = run<KIND, 64, MODE>(iters);
    // instructions of the kind issued per SIMD and iteration: (waves per SIMD running them) x NV
    const double wv = MODE == 0 ? 4.0 : 2.0;
    auto cyc = [&](float t, int nv) { return (double)(t - t0) * 1e-3 * ghz * 1e9 / ((double)iters * nv * wv); };
    printf("mode %d %-28s base %.3f ms | +8: %.3f ms (%.1f cyc/inst) | +32: %.3f ms (%.1f) | +64: %.3f ms (%.1f)\n", MODE, name, t0, t8, cyc(t8, 8), t32,
           cyc(t32, 32), t64, cyc(t64, 64));
}

int main()
{
    hipMalloc(&g_out, sizeof(float) * 256 * 1024);
    const int iters = 2000;
    const double ghz = 2.38;
    // base: 4 waves x 32 MFMA x 64 cycles per iteration and SIMD = 8192 cycles
    sweep<0, 0>("v_add_f32", iters, ghz);
    sweep<1, 0>("v_add_u32", iters, ghz);
    sweep<2, 0>("v_max_f32_dpp row_shr", iters, ghz);
    sweep<3, 0>("ds_read_b32", iters, ghz);
    sweep<4, 0>("s_add_u32", iters, ghz);
    sweep<5, 0>("v_fma_f32", iters, ghz);
    sweep<6, 0>("v_and_b32", iters, ghz);
    sweep<7, 0>("v_mov_b32", iters, ghz);
    sweep<8, 0>("v_lshl_add_u32", iters, ghz);
    sweep<9, 0>("v_cmp+v_cndmask (2 inst)", iters, ghz);
    sweep<10, 0>("v_readfirstlane_b32", iters, ghz);
    sweep<0, 1>("v_add_f32", iters, ghz);
    sweep<1, 1>("v_add_u32", iters, ghz);
    sweep<3, 1>("ds_read_b32", iters, ghz);
    sweep<4, 1>("s_add_u32", iters, ghz);
    sweep<7, 1>("v_mov_b32", iters, ghz);
    return 0;
}
