// tools/serial_stretch_probe.hip -- how much slower does SERIAL code (dependent vector ALU chains, LDS read -> use chains)
// run on a SIMD whose three other waves issue v_mfma_f32_32x32x2_f32 back to back, and what does s_setprio change?
// One workgroup of 16 waves per CU: waves 0-3 (one per SIMD) run the serial chain and time it with s_memtime; waves 4-15 run
// MFMAs until the serial waves are done (LDS flag).  Build: hipcc -O3 --offload-arch=gfx950 -o tools/serial_stretch_probe ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: dependent v_add_f32 chain; 1: ds_read_b32 -> v_add_f32 -> address chain; 2: independent v_add_f32 x4 interleaved
// 3: v_mfma 4x4x1 (short matrix op) chain as the "serial" work
template <int KIND>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, int reps, int mfma_on, int prio, int mfma_prio, const unsigned* zeros)
{
    __shared__ float buf[2048];
    __shared__ int done;
    buf[threadIdx.x] = 0.0f; buf[threadIdx.x + 1024] = 0.0f;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
        float x = (float)threadIdx.x, y = 1.0f, x1 = x, x2 = x, x3 = x;
        unsigned addr = (threadIdx.x & 63) * 4;
        const unsigned* gp = zeros + (threadIdx.x & 63);
        const unsigned* sp = zeros;
        __builtin_amdgcn_s_sleep(64);                       // let the matrix waves get going
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; ++r) {
            if constexpr (KIND == 0) {
#pragma unroll
                for (int u = 0; u < 64; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
            } else if constexpr (KIND == 1) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    float v;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(v));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                }
            } else if constexpr (KIND == 4) {             // LDS round trips only
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    unsigned v;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
                    addr = (addr + v) & 0xfff;            // address dependence through an integer add (one vector op)
                }
            } else if constexpr (KIND == 5) {             // 8 independent LDS reads, one wait
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    unsigned v0, v1, v2, v3, v4, v5, v6, v7;
                    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:256\n\tds_read_b32 %2, %8 offset:512\n\tds_read_b32 %3, %8 offset:768\n\t"
                                 "ds_read_b32 %4, %8 offset:1024\n\tds_read_b32 %5, %8 offset:1280\n\tds_read_b32 %6, %8 offset:1536\n\tds_read_b32 %7, %8 offset:1792\n\t"
                                 "s_waitcnt lgkmcnt(0)" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr) : "memory");
                    addr = (addr + (v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7)) & 0x7ff;
                }
            } else if constexpr (KIND == 6) {             // global (L2-resident) round trips
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned v;
                    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(gp) : "memory");
                    gp += v;
                }
            } else if constexpr (KIND == 7) {             // scalar loads (constant cache) round trips
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    unsigned v;
                    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(sp) : "memory");
                    sp += v;
                }
            } else if constexpr (KIND == 8) {             // LDS write -> read back (what a broadcast through LDS costs)
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    unsigned v;
                    asm volatile("ds_write_b32 %1, %2\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "v"(addr) : "memory");
                    addr = (addr + (v & 4)) & 0xfff;
                }
            } else if constexpr (KIND == 2) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(y));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(y));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(y));
                }
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 4 + wave] = t1 - t0; atomicAdd(&done, 1); }
        buf[threadIdx.x] = x + x1 + x2 + x3;
    } else if (wave < 4 + 4 * mfma_on) {                     // mfma_on = matrix waves per SIMD (0..3)
        (void)mfma_prio;
        f32x16 acc0, acc1;
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        const float a = threadIdx.x * 1e-3f, b = threadIdx.x * 1e-4f;
        while (*(volatile int*)&done < 4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
        buf[threadIdx.x] = s;
    }
}

static unsigned* g_zeros;
template <int KIND> void run(const char* name, int ops_per_rep, unsigned long long* dout)
{
    const int reps = 100, nb = 256;
    double base = 0;
    for (int m = 0; m <= 3; ++m) {
        hipLaunchKernelGGL((probe<KIND>), dim3(nb), dim3(1024), 0, 0, dout, reps, m, 0, 0, g_zeros);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nb * 4);
        hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double t = (double)h[h.size() / 2] / ((double)reps * ops_per_rep);
        if (m == 0) base = t;
        printf("%-40s %d matrix waves on the SIMD: median %9.1f ticks/op  (x%.1f)\n", name, m, t, t / base);
    }
}

int main()
{
    unsigned long long* dout; hipMalloc(&dout, 8 * 1024);
    hipMalloc(&g_zeros, 4096); hipMemset(g_zeros, 0, 4096);
    run<0>("dependent v_add_f32 chain", 64, dout);
    run<2>("4 independent v_add_f32 chains", 64, dout);
    run<4>("ds_read -> wait -> 1 int op", 16, dout);
    run<5>("8 independent ds_read -> wait", 2, dout);
    run<8>("ds_write + ds_read same address -> wait", 16, dout);
    run<6>("global_load (L2) -> wait", 4, dout);
    run<7>("s_load_dword -> wait", 16, dout);
    run<1>("ds_read -> 3 dependent v_add", 16, dout);
    return 0;
}
