// hscmp_mfma.h -- matrix-core (MFMA) variants of the two correlation kernels, float32, gfx950.
//
// The dictionary-vs-residual correlation c[t,k] = sum_w r[t-off+w] * D[k,w] is a Toeplitz
// contraction: a [32 atoms] x [32 positions] output tile is 32 chained
// v_mfma_f32_32x32x2_f32 (for W=64), each adding taps (2s, 2s+1):
//     A[i][kk] = D[atom0+i][2s+kk]                 (one VGPR: lane l -> i = l&31, kk = l>>5)
//     B[kk][j] = r[pos0+j - off + 2s+kk]           (one VGPR: lane l -> j = l&31, kk = l>>5)
// The f32 MFMA is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md "FP32-input
// MFMA"), and the taps enter in ascending order, so every c[t,k] equals the oracle's pinned
// sequential chain exactly -- parity is bit-exact, not approximate.
//
// Atoms sit on the ROWS of the accumulator tile (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)),
// positions on its columns (col = lane&31): a lane holds 16 atoms of ONE position, so the
// arg-max over atoms (per-position best, table-free state) is a lane-local compare chain plus
// one exchange between the two half-waves -- no cross-lane reduction tree, no table in HBM.
//
// LDS images:
//   dictionary  Dimg[g][s4][lane][4]  : group g of 32 atoms, chunk s4 of 4 k-steps; one
//               ds_read_b128 per lane fetches the A operands of 4 consecutive MFMAs, lane-linear
//               (conflict-free).  Zero padded to 32-atom groups and 8-tap chunks: a zero tap
//               leaves the chain unchanged (fma(x, 0, acc) == acc).
//   signal      plain floats; the B operand of k-step s for lane (j,kk) is win[j + kk + 2s]:
//               stride-1 across lanes (conflict-free ds_read_b32).
#pragma once

#include "hscmp_kernels.h"

#include <vector>

namespace hscmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaMaxSeg = 512;          // smaller control block: leaves LDS for the dictionary image
constexpr int kMfmaChunk = 2048;          // positions per workgroup of the initial correlation
constexpr size_t kMfmaMaxDictBytes = 64 * 1024;

struct MfmaArgs {
    const float* dimg;   // device dictionary image
    int G;               // atom groups of 32
    int S4;              // chunks of 4 k-steps (8 taps)
    int has_w;
};

inline int mfma_groups(int K) { return (K + 31) / 32; }
inline int mfma_chunks(int W) { return (W + 7) / 8; }

inline bool mfma_supported(int K, int W, int F)
{
    if (F != 1) return false;
    const size_t bytes = (size_t)mfma_groups(K) * mfma_chunks(W) * 64 * 4 * sizeof(float);
    return bytes <= kMfmaMaxDictBytes && W <= 128;
}

// host: Dimg[g][s4][lane][q] = D[32g + (lane&31)][2*(4*s4+q) + (lane>>5)], zero padded
inline void mfma_build_dict_image(const float* D, int K, int W, int F, std::vector<float>& out)
{
    (void)F;
    const int G = mfma_groups(K), S4 = mfma_chunks(W);
    out.assign((size_t)G * S4 * 64 * 4, 0.0f);
    for (int g = 0; g < G; ++g)
        for (int s4 = 0; s4 < S4; ++s4)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 4; ++q) {
                    const int k = 32 * g + (lane & 31);
                    const int w = 2 * (4 * s4 + q) + (lane >> 5);
                    if (k < K && w < W) out[(((size_t)g * S4 + s4) * 64 + lane) * 4 + q] = D[(size_t)k * W + w];
                }
}

// ------------------------------------------------------------------------------------------------
// One 32-position tile against all atom groups: per-position best (coefficient, atom).
//   dimg : LDS dictionary image;  win : LDS floats, win[j + kk + 2s] is the B operand (see above)
//   wts  : LDS weights [32*G] (HAS_W) ;  result valid in lanes 0..31 (position = lane)
// S4C > 0: compile-time chunk count (B operands live in registers); S4C == 0: runtime count.
// ------------------------------------------------------------------------------------------------
template <int S4C, bool HAS_W>
__device__ __forceinline__ void mfma_tile_best(const float* __restrict__ dimg, const float* __restrict__ win,
                                               const float* __restrict__ wts, int G, int S4rt, int lane,
                                               float& out_c, int& out_k)
{
    const int S4 = S4C > 0 ? S4C : S4rt;
    const int j = lane & 31, h = lane >> 5;
    const float* wb = win + j + h;
    float bs = -1.0f, bc = 0.0f;
    int bk = 0;

    float bop[S4C > 0 ? 4 * S4C : 1];
    if (S4C > 0) {
#pragma unroll
        for (int s = 0; s < 4 * S4C; ++s) bop[s] = wb[2 * s];
    }

    const f32x4* dv = reinterpret_cast<const f32x4*>(dimg) + lane;
    for (int g = 0; g < G; ++g) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        if (S4C > 0) {
#pragma unroll
            for (int s4 = 0; s4 < S4C; ++s4) {
                const f32x4 a = dv[(g * S4C + s4) * 64];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bop[4 * s4 + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bop[4 * s4 + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], bop[4 * s4 + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], bop[4 * s4 + 3], acc, 0, 0, 0);
            }
        } else {
            for (int s4 = 0; s4 < S4; ++s4) {
                const f32x4 a = dv[(g * S4 + s4) * 64];
                const float b0 = wb[8 * s4 + 0], b1 = wb[8 * s4 + 2], b2 = wb[8 * s4 + 4], b3 = wb[8 * s4 + 6];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b3, acc, 0, 0, 0);
            }
        }
        // lane-local arg-max over this lane's 16 atoms, ascending atom index, strict > (first k wins)
        const int kbase = 32 * g + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kbase + (r & 3) + 8 * (r >> 2);
            const float v = acc[r];
            float s;
            if (HAS_W) { const float sw = v * wts[k]; s = fabsf(sw); } else s = fabsf(v);
            if (s > bs) { bs = s; bc = v; bk = k; }
        }
    }
    // merge the two half-waves (same position, interleaved atom sets): larger score, then lower k
    const float os = __shfl_xor(bs, 32);
    const float oc = __shfl_xor(bc, 32);
    const int ok = __shfl_xor(bk, 32);
    if (os > bs || (os == bs && ok < bk)) { bc = oc; bk = ok; }
    out_c = bc;
    out_k = bk;
}

__device__ __forceinline__ void lds_copy_f32(float* dst, const float* __restrict__ src, int n)
{
    // n is a multiple of 4; 16-byte coalesced copy
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    for (int i = threadIdx.x; i < n / 4; i += kThreads) d4[i] = s4[i];
}

// ------------------------------------------------------------------------------------------------
// initial correlation (modeling.py:1077), zero-padded 'same', reduced to the per-position best.
//   grid = (ceil(T / kMfmaChunk), B), block = kThreads; each wave owns every 4th 32-position tile
// LDS: [dictionary image][weights 32*G][signal chunk + halo]
// ------------------------------------------------------------------------------------------------
template <int S4C, bool HAS_W>
__global__ __launch_bounds__(kThreads) void corr_init_mfma_kernel(DevParams P, State<float> S, MfmaArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = A.G, S4 = S4C > 0 ? S4C : A.S4;
    const int nd = G * S4 * 256;                        // floats in the dictionary image
    float* dimg = reinterpret_cast<float*>(smem);
    float* wts = dimg + nd;
    float* xs = wts + 32 * G;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = P.T;
    const int c0 = blockIdx.x * kMfmaChunk;
    const int npos = min(kMfmaChunk, T - c0);
    const int nx = kMfmaChunk + 8 * S4 + 32;            // chunk + taps + slack for the last tile's kk offset

    lds_copy_f32(dimg, A.dimg, nd);
    if (HAS_W) for (int i = tid; i < 32 * G; i += kThreads) wts[i] = i < P.K ? S.weights[i] : 0.0f;
    const float* x = S.residual + (int64_t)b * T;       // residual == copy of the signal at this point
    for (int i = tid; i < nx; i += kThreads) {
        const int g = c0 - P.off + i;                   // zero padding of 'same' (modeling.py:159-164)
        xs[i] = (g >= 0 && g < T) ? x[g] : 0.0f;
    }
    __syncthreads();

    const int ntiles = (npos + 31) / 32;
    for (int q = wv; q < ntiles; q += kWaves) {
        float c; int k;
        mfma_tile_best<S4C, HAS_W>(dimg, xs + 32 * q, wts, G, S4, lane, c, k);
        const int t = c0 + 32 * q + lane;
        if (lane < 32 && t < T) {
            S.best_c[(int64_t)b * T + t] = c;
            S.best_k[(int64_t)b * T + t] = k;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MfmaRecorr: policy of iterate_kernel -- re-correlation of the 2W-1 touched rows on the matrix
// cores.  Policy LDS: [dictionary image][weights][reflect-padded residual window]
// ------------------------------------------------------------------------------------------------
template <int S4C, bool HAS_W> struct MfmaRecorr {
    static constexpr int kMaxSegments = kMfmaMaxSeg;
    using Shared = IterSharedT<float, kMfmaMaxSeg>;
    using Args = MfmaArgs;

    static int window_floats(int W, int S4) { return ((2 * W - 1 + 31) / 32) * 32 + 8 * S4 + 32; }
    static size_t extra_lds_bytes(const DevParams& P, const Args& A)
    {
        return ((size_t)A.G * A.S4 * 256 + 32 * A.G + window_floats(P.W, A.S4)) * sizeof(float);
    }

    static __device__ __forceinline__ void prologue(const DevParams& P, const State<float>& S, const Args& A, char* lds)
    {
        const int S4 = S4C > 0 ? S4C : A.S4;
        const int nd = A.G * S4 * 256;
        float* dimg = reinterpret_cast<float*>(lds);
        float* wts = dimg + nd;
        lds_copy_f32(dimg, A.dimg, nd);
        if (HAS_W) for (int i = threadIdx.x; i < 32 * A.G; i += kThreads) wts[i] = i < P.K ? S.weights[i] : 0.0f;
        // visibility: the caller's next __syncthreads()
    }

    template <typename SH>
    static __device__ __forceinline__ void run(const DevParams& P, const State<float>&, const Sig<float>& Gs,
                                               SH&, const Args& A, char* lds, int p)
    {
        const int T = P.T, W = P.W, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const int S4 = S4C > 0 ? S4C : A.S4;
        const int nd = A.G * S4 * 256;
        float* dimg = reinterpret_cast<float*>(lds);
        float* wts = dimg + nd;
        float* win = wts + 32 * A.G;
        const int nrows = 2 * W - 1;
        const int ntiles = (nrows + 31) / 32;
        const int nwin = ntiles * 32 + 8 * S4 + 32;
        const int tstart = p - P.off - (W - 1);            // modeling.py:1028-1033
        const int tend = p + W / 2 + (W - 1);              // :1038
        const int sidx = tstart < 0 ? 0 : tstart;          // :1034
        const int eidx = tend > T - 1 ? T - 1 : tend;      // :1039
        const int nslice = eidx - sidx + 1;
        const int span = 3 * W - 2;
        // reflect-padded residual span (np.pad mode='reflect', :1046); zeros behind it feed only
        // zero taps / rows that are never written
        for (int i = tid; i < nwin; i += kThreads) {
            float v = 0.0f;
            if (i < span) v = Gs.r[reflect_index(tstart + i, sidx, nslice)];
            win[i] = v;
        }
        __syncthreads();
        for (int q = wv; q < ntiles; q += kWaves) {
            float c; int k;
            mfma_tile_best<S4C, HAS_W>(dimg, win + 32 * q, wts, A.G, S4, lane, c, k);
            const int row = 32 * q + lane;
            const int t = p - (W - 1) + row;
            if (lane < 32 && row < nrows && t >= 0 && t < T) {      // overlapReplace clipping (utils.py:133-161)
                Gs.bc[t] = c;
                Gs.bk[t] = k;
            }
        }
    }
};

// host-side dispatch -----------------------------------------------------------------------------
template <int S4C, bool HAS_W>
static int mfma_launch_corr_init_t(hipStream_t stream, const DevParams& P, const State<float>& S, const MfmaArgs& A)
{
    const size_t lds = ((size_t)A.G * A.S4 * 256 + 32 * A.G + kMfmaChunk + 8 * A.S4 + 32) * sizeof(float);
    auto kern = corr_init_mfma_kernel<S4C, HAS_W>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    dim3 grid((P.T + kMfmaChunk - 1) / kMfmaChunk, P.B);
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, stream, P, S, A);
    return 0;
}

template <int S4C, bool HAS_W>
static int mfma_launch_iterate_t(hipStream_t stream, const DevParams& P0, const State<float>& S, const MfmaArgs& A)
{
    using Pol = MfmaRecorr<S4C, HAS_W>;
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    const size_t lds = ((sizeof(typename Pol::Shared) + 15) / 16) * 16 + Pol::extra_lds_bytes(P, A);
    auto kern = iterate_kernel<float, Pol>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(kern, dim3(P.B), dim3(kThreads), lds, stream, P, S, A);
    return 0;
}

inline MfmaArgs mfma_args(const DevParams& P, const State<float>& S, const float* dimg)
{
    MfmaArgs A;
    A.dimg = dimg; A.G = mfma_groups(P.K); A.S4 = mfma_chunks(P.W); A.has_w = S.weights != nullptr;
    return A;
}

#define HSCMP_MFMA_DISPATCH(FN)                                                         \
    do {                                                                                \
        const bool hw = A.has_w != 0;                                                   \
        switch (A.S4) {                                                                 \
        case 8: return hw ? FN<8, true>(stream, P, S, A) : FN<8, false>(stream, P, S, A);   \
        case 4: return hw ? FN<4, true>(stream, P, S, A) : FN<4, false>(stream, P, S, A);   \
        case 2: return hw ? FN<2, true>(stream, P, S, A) : FN<2, false>(stream, P, S, A);   \
        default: return hw ? FN<0, true>(stream, P, S, A) : FN<0, false>(stream, P, S, A);  \
        }                                                                               \
    } while (0)

inline int mfma_launch_corr_init(hipStream_t stream, const DevParams& P, const State<float>& S, const float* dimg)
{
    const MfmaArgs A = mfma_args(P, S, dimg);
    HSCMP_MFMA_DISPATCH(mfma_launch_corr_init_t);
}

inline int mfma_launch_iterate(hipStream_t stream, const DevParams& P, const State<float>& S, const float* dimg)
{
    const MfmaArgs A = mfma_args(P, S, dimg);
    HSCMP_MFMA_DISPATCH(mfma_launch_iterate_t);
}

}  // namespace hscmp
