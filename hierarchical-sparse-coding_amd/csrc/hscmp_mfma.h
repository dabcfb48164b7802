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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace hscmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int kMfmaMaxSeg = 512;          // smaller control block: leaves LDS for the dictionary image
constexpr int kMfmaChunk = 2048;          // positions per workgroup of the initial correlation

template <typename R> struct MfmaArgsT {
    const R* dimg;       // device dictionary image
    int G;               // atom groups (32 atoms for float32, 16 for float64)
    int S4;              // chunks of 8 taps (one 16-byte A-operand word per lane)
    int has_w;
};
using MfmaArgs = MfmaArgsT<float>;

inline int mfma_groups(int K) { return (K + 31) / 32; }
inline int mfma_chunks(int W) { return (W + 7) / 8; }

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
// S4C > 0: compile-time chunk count -- B operands live in registers, A operands and accumulators
// are double buffered: group g+1's A fragments are fetched while group g's MFMA chain runs, and
// the lane-local arg-max of group g-1 is interleaved between the MFMA issues of group g.
// S4C == 0: runtime chunk count (any W), simple loop.
// ------------------------------------------------------------------------------------------------
// value of the same lane index in the OTHER half-wave (lane ^ 32) through v_permlane32_swap:
// swap(a, a) leaves r[0] = {low half of a, low half of a} and r[1] = {high half, high half}
__device__ __forceinline__ int swap_halves_i(int v, int h)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return h ? (int)r[0] : (int)r[1];
}
__device__ __forceinline__ float swap_halves_f(float v, int h) { return __int_as_float(swap_halves_i(__float_as_int(v), h)); }

// ------------------------------------------------------------------------------------------------
// One 32-position tile against all atom groups: per-position best SCORE max_k |c[t,k] * w_k|.
//   dimg : LDS dictionary image;  win : LDS floats, win[j + kk + 2s] is the B operand (see above)
//   wts  : LDS weights [32*G] (HAS_W) ;  result valid in every lane (position = lane & 31)
//
// Score-only state: the fp32 MFMA shares the FP32 datapath with the vector ALU (measured: VALU
// work next to the MFMA chain does not overlap, it adds), so the per-element cost of the arg-max
// is what limits the kernel.  Keeping only the score needs one v_max3_f32 per TWO accumulator
// elements (|.| is a free source modifier) instead of compare + two selects per element; the atom
// index and the signed coefficient of a position are recomputed ("resolved") from one row of
// K*W multiply-adds only when that position is actually selected.
//
// S4C > 0: compile-time chunk count -- B operands live in registers, A operands and accumulators
// are double buffered: group g+1's A fragments are fetched while group g's MFMA chain runs, and
// the reduction of group g-1 is interleaved between the MFMA issues of group g.
// S4C == 0: runtime chunk count (any W), simple loop.
// ------------------------------------------------------------------------------------------------
template <bool HAS_W>
__device__ __forceinline__ void mfma_reduce_pair(float v0, float v1, int k0, int k1, const float* __restrict__ wts, float& bs)
{
#ifdef HSCMP_DBG_NO_REDUCE   // diagnostic build: keep the accumulator live, skip the reduction
    asm volatile("" :: "v"(v0), "v"(v1));
    (void)k0; (void)k1; (void)wts; (void)bs;
#else
    if (HAS_W) { v0 = v0 * wts[k0]; v1 = v1 * wts[k1]; }      // modeling.py:906, one rounded product each
    bs = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), bs);              // v_max3_f32 |v0|, |v1|, bs
#endif
}

// The group hint: next to the score the tile reports WHICH 32-atom group holds the first atom that attains it (two
// vector instructions per group: did the running maximum grow while this group was reduced?).  When the position is
// selected, a wave recomputes that group's rows only -- 32 chains, one per lane, instead of K spread over the
// workgroup -- and needs nobody else's result (see resolve_group).
__device__ __forceinline__ void mfma_merge_halves(float& bs, int& bg, int h)
{
    // the two half-waves hold the same position with interleaved atom sets: larger score, on a tie the lower group
    const float os = swap_halves_f(bs, h);
    const int og = swap_halves_i(bg, h);
    bg = (os > bs || (os == bs && og < bg)) ? og : bg;
    bs = fmaxf(bs, os);
}

template <int S4C, bool HAS_W>
__device__ __forceinline__ float mfma_tile_score(const float* __restrict__ dimg, const float* __restrict__ win,
                                                 const float* __restrict__ wts, int G, int S4rt, int lane, int& grp)
{
    const int j = lane & 31, h = lane >> 5;
    const float* wb = win + j + h;
    float bs = 0.0f;                           // scores are >= 0
    int bg = 0;
    const f32x4* dv = reinterpret_cast<const f32x4*>(dimg) + lane;
    // atom of accumulator element r: 32g + 4h + (r&3) + 8(r>>2); pair (2e, 2e+1) shares one v_max3
    auto katom = [&](int kbase, int r) { return kbase + (r & 3) + 8 * (r >> 2); };

    if constexpr (S4C > 0) {
        constexpr int NM = 4 * S4C;            // MFMAs per atom group
        float bop[NM];
#pragma unroll
        for (int s = 0; s < NM; ++s) bop[s] = wb[2 * s];
        f32x4 a0[S4C], a1[S4C];
        f32x16 acc0, acc1;

        auto load_a = [&](f32x4 (&a)[S4C], int g) {
#ifdef HSCMP_DBG_NO_AREAD    // diagnostic build: A operands from registers, no LDS reads
#pragma unroll
            for (int s4 = 0; s4 < S4C; ++s4) { a[s4][0] = bop[s4]; a[s4][1] = bop[s4 + 1]; a[s4][2] = (float)g; a[s4][3] = bop[0]; }
#else
#pragma unroll
            for (int s4 = 0; s4 < S4C; ++s4) a[s4] = dv[(g * S4C + s4) * 64];
#endif
        };
        auto run_first = [&](const f32x4 (&a)[S4C], f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m >> 2][m & 3], bop[m], acc, 0, 0, 0);
        };
        // MFMA chain of one group into `acc`; the 8 reduction steps of the previous group's
        // accumulator `accp` are spread between the MFMA issues
        auto run_next = [&](const f32x4 (&a)[S4C], f32x16& acc, const f32x16& accp, int gp) {
            const int kbasep = 32 * gp + 4 * h;
            const float before = bs;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m >> 2][m & 3], bop[m], acc, 0, 0, 0);
#pragma unroll
                for (int e = (m * 8) / NM; e < ((m + 1) * 8) / NM; ++e)
                    mfma_reduce_pair<HAS_W>(accp[2 * e], accp[2 * e + 1], katom(kbasep, 2 * e), katom(kbasep, 2 * e + 1), wts, bs);
            }
            bg = bs > before ? gp : bg;
        };
        auto reduce_all = [&](const f32x16& acc, int gl) {
            const int kbase = 32 * gl + 4 * h;
            const float before = bs;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                mfma_reduce_pair<HAS_W>(acc[2 * e], acc[2 * e + 1], katom(kbase, 2 * e), katom(kbase, 2 * e + 1), wts, bs);
            bg = bs > before ? gl : bg;
        };

        load_a(a0, 0);
        if (G > 1) load_a(a1, 1);
        run_first(a0, acc0);
        int g = 1;
        for (; g + 1 < G; g += 2) {
            load_a(a0, g + 1);
            run_next(a1, acc1, acc0, g - 1);
            if (g + 2 < G) load_a(a1, g + 2);
            run_next(a0, acc0, acc1, g);
        }
        if (g < G) {
            run_next(a1, acc1, acc0, g - 1);
            reduce_all(acc1, g);
        } else {
            reduce_all(acc0, G - 1);
        }
    } else {
        const int S4 = S4rt;
        for (int g = 0; g < G; ++g) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            for (int s4 = 0; s4 < S4; ++s4) {
                const f32x4 a = dv[(g * S4 + s4) * 64];
                const float b0 = wb[8 * s4 + 0], b1 = wb[8 * s4 + 2], b2 = wb[8 * s4 + 4], b3 = wb[8 * s4 + 6];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b3, acc, 0, 0, 0);
            }
            const int kbase = 32 * g + 4 * h;
            const float before = bs;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                mfma_reduce_pair<HAS_W>(acc[2 * e], acc[2 * e + 1], katom(kbase, 2 * e), katom(kbase, 2 * e + 1), wts, bs);
            bg = bs > before ? g : bg;
        }
    }
    mfma_merge_halves(bs, bg, h);
    grp = bg;
    return bs;
}

// The same tile for a kernel that runs FOUR waves per SIMD (128 VGPRs per lane; MfmaRecorr with four signals per
// workgroup).  Neither the 32 B operands nor a group's 32 A operands stay in registers: both stream from LDS one chunk
// (4 MFMAs) ahead of their use -- no MFMA waits for a read issued right in front of it: in the greedy loop fewer than
// four waves of a SIMD are inside a tile at any time, and what one wave waits for nobody covers.  Two accumulators alternate, so the 8
// v_max3 of group g-1 sit between the MFMAs of group g.  Nothing is scheduled across a chunk boundary (left alone the
// scheduler hoists the operand reads of several chunks, the tile then wants 145 VGPRs by itself and the kernel around
// it spills).  Same products, same order: bit-identical to mfma_tile_score.
template <int S4C, bool HAS_W>
__device__ __forceinline__ float mfma_tile_score_lean(const float* __restrict__ dimg, const float* __restrict__ win,
                                                      const float* __restrict__ wts, int G, int lane, int& grp)
{
    static_assert(S4C > 0, "compile-time chunk count only");
    const int j = lane & 31, h = lane >> 5;
    const float* wb = win + j + h;
    const f32x4* dv = reinterpret_cast<const f32x4*>(dimg) + lane;
    float bs = 0.0f;
    int bg = 0;
    auto katom = [&](int kbase, int r) { return kbase + (r & 3) + 8 * (r >> 2); };
    const int kb0 = 4 * h;
    f32x4 a = dv[0];                                            // chunk 0 of group 0
    float b0 = wb[0], b1 = wb[2], b2 = wb[4], b3 = wb[6];
    // MFMA chain of group g into `acc`; with REDUCE the 8 reduction steps of the previous group's finished
    // accumulator `accp` sit between the MFMAs (first MFMA of a chain: C = 0).  Leaves `a` = chunk 0 of group g+1.
    auto chain = [&](int g, f32x16& acc, const f32x16& accp, auto reduce_tag) {
        constexpr bool REDUCE = decltype(reduce_tag)::value;
        const float before = bs;
#pragma unroll
        for (int s4 = 0; s4 < S4C; ++s4) {
            // the next chunk's A operands (the next group's first chunk behind the last one; one chunk past the image
            // for the last group: inside the LDS allocation, never used), in flight under this chunk's MFMAs
            const f32x4 an = dv[(g * S4C + s4 + 1) * 64];
            const int sn = (s4 + 1) % S4C;                      // B operands: the same 4 * S4C values for every group
            const float bn0 = wb[8 * sn + 0], bn1 = wb[8 * sn + 2], bn2 = wb[8 * sn + 4], bn3 = wb[8 * sn + 6];
            if (s4 == 0) {
                f32x16 zero;
#pragma unroll
                for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b0, zero, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b0, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b1, acc, 0, 0, 0);
            if constexpr (REDUCE) {
#pragma unroll
                for (int e = (s4 * 8) / S4C; e < ((s4 + 1) * 8) / S4C; ++e)
                    mfma_reduce_pair<HAS_W>(accp[2 * e], accp[2 * e + 1], katom(32 * (g - 1) + kb0, 2 * e), katom(32 * (g - 1) + kb0, 2 * e + 1), wts, bs);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b3, acc, 0, 0, 0);
            a = an; b0 = bn0; b1 = bn1; b2 = bn2; b3 = bn3;
#ifndef HSCMP_LEAN_NO_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if constexpr (REDUCE) bg = bs > before ? g - 1 : bg;
    };
    auto reduce_all = [&](const f32x16& acc, int g) {
        const float before = bs;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            mfma_reduce_pair<HAS_W>(acc[2 * e], acc[2 * e + 1], katom(32 * g + kb0, 2 * e), katom(32 * g + kb0, 2 * e + 1), wts, bs);
        bg = bs > before ? g : bg;
    };
    f32x16 acc0, acc1;
    chain(0, acc0, acc0, std::false_type());
    int g = 1;
#pragma unroll 1
    for (; g + 1 < G; g += 2) {
        chain(g, acc1, acc0, std::true_type());
        chain(g + 1, acc0, acc1, std::true_type());
    }
    if (g < G) {
        chain(g, acc1, acc0, std::true_type());
        reduce_all(acc1, g);
    } else {
        reduce_all(acc0, G - 1);
    }
    mfma_merge_halves(bs, bg, h);
    grp = bg;
    return bs;
}

__device__ __forceinline__ void lds_copy16(void* dst, const void* __restrict__ src, int nbytes, int tid = (int)threadIdx.x, int nthreads = kThreads)
{
    // nbytes is a multiple of 16; coalesced copy, 8 loads in flight per thread before the stores
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    const int n4 = nbytes / 16;
    for (int base = 0; base < n4; base += 8 * nthreads) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nthreads + tid;
            if (i < n4) v[u] = s4[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nthreads + tid;
            if (i < n4) d4[i] = v[u];
        }
    }
}

// base of the kernel's dynamic LDS array (policies whose signals share a region in front of their own)
__device__ __forceinline__ char* dyn_lds()
{
    extern __shared__ __attribute__((aligned(16))) char hscmp_dyn_lds_[];
    return hscmp_dyn_lds_;
}

// ------------------------------------------------------------------------------------------------
// initial correlation (modeling.py:1077), zero-padded 'same', reduced to the per-position best.
//   PERSISTENT grid: gridDim.x workgroups (2 per CU) walk the (signal, chunk) items; the 64 KB
//   dictionary image is loaded into LDS once per workgroup, the next chunk of the signal is
//   prefetched into registers while the current one is on the matrix cores.
//   Each wave owns every 4th 32-position tile of a chunk.
// LDS: [dictionary image][weights 32*G][signal chunk + halo]
// ------------------------------------------------------------------------------------------------
constexpr int kMfmaChunkLoads = (kMfmaChunk + 128 + 32 + kThreads - 1) / kThreads;   // chunk + max taps (W<=128) + slack

template <typename Tile, int S4C, bool HAS_W>
__global__ __launch_bounds__(kThreads) void corr_init_mfma_kernel(DevParams P, State<typename Tile::R> S, MfmaArgsT<typename Tile::R> A)
{
    using R = typename Tile::R;
    constexpr int TP = Tile::TP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = A.G, S4 = S4C > 0 ? S4C : A.S4;
    const int nd = G * S4 * Tile::kChunkElems;          // elements of the dictionary image
    R* dimg = reinterpret_cast<R*>(smem);
    R* wts = dimg + nd;
    R* xs = wts + Tile::GA * G;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = P.T;
    const int nx = kMfmaChunk + 8 * S4 + 32;            // chunk + taps + slack for the last tile's kk offset
    const int cps = (T + kMfmaChunk - 1) / kMfmaChunk;  // chunks per signal
    const int nitems = cps * P.B;

    lds_copy16(dimg, A.dimg, nd * (int)sizeof(R));
    if (HAS_W) for (int i = tid; i < Tile::GA * G; i += kThreads) wts[i] = i < P.K ? S.weights[i] : (R)0;

    R xr[kMfmaChunkLoads];
    auto fetch = [&](int item) {                        // global -> registers (zero padding of 'same', :159-164)
        const int b = item / cps, c0 = (item % cps) * kMfmaChunk;
        const R* x = S.residual + (int64_t)b * T;       // residual == copy of the signal at this point
#pragma unroll
        for (int u = 0; u < kMfmaChunkLoads; ++u) {
            const int i = u * kThreads + tid;
            const int g = c0 - P.off + i;
            xr[u] = (i < nx && g >= 0 && g < T) ? x[g] : (R)0;
        }
    };
    auto stash = [&]() {                                // registers -> LDS
#pragma unroll
        for (int u = 0; u < kMfmaChunkLoads; ++u) {
            const int i = u * kThreads + tid;
            if (i < nx) xs[i] = xr[u];
        }
    };

    int item = blockIdx.x;
    if (item < nitems) fetch(item);
    for (; item < nitems; item += gridDim.x) {
        stash();
        __syncthreads();
        const int next = item + gridDim.x;
        if (next < nitems) fetch(next);                 // in flight while this chunk is computed
        const int b = item / cps, c0 = (item % cps) * kMfmaChunk;
        const int npos = min(kMfmaChunk, T - c0);
        const int ntiles = (npos + TP - 1) / TP;
        for (int q = wv; q < ntiles; q += kWaves) {
            int grp;
            const R sc = Tile::template tile_score<S4C, HAS_W>(dimg, xs + TP * q, wts, G, S4, lane, grp);
            const int t = c0 + TP * q + lane;
            if (lane < TP && t < T) {                                       // score-only state (see mfma_tile_score)
                S.best_c[(int64_t)b * T + t] = sc;
                S.best_k[(int64_t)b * T + t] = grp;                         // the group hint, not an atom
            }
        }
        __syncthreads();                                // all tiles read xs before it is overwritten
    }
}

// value held by lane `src` (wave-uniform index) in every lane: v_readlane, no LDS crossbar
__device__ __forceinline__ float wave_bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double wave_bcast(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (long long)lo);
}

__device__ __forceinline__ int dimg_index(int k, int w, int S4)
{
    // float index of D[k][w] inside Dimg[g][s4][lane][q]  (see mfma_build_dict_image)
    return (((k >> 5) * S4 + (w >> 3)) * 64 + (k & 31) + 32 * (w & 1)) * 4 + ((w >> 1) & 3);
}

// D[k][.] . rwin[.] as the pinned sequential fma chain (taps ascending), D read from the LDS image:
// for chunk s4 the two 16-byte words of lanes (k&31) and (k&31)+32 hold taps 8*s4 + {0,2,4,6} and
// 8*s4 + {1,3,5,7}.  fma(r, d, acc) == fma(d, r, acc): bit-identical to the MFMA chain.
template <int S4C>
__device__ __forceinline__ float resolve_chain(const float* __restrict__ dimg, const float* __restrict__ rwin, int k, int S4rt)
{
    const int S4 = S4C > 0 ? S4C : S4rt;
    const f32x4* dv = reinterpret_cast<const f32x4*>(dimg) + ((k >> 5) * S4) * 64 + (k & 31);
    const f32x4* rw = reinterpret_cast<const f32x4*>(rwin);
    float acc = 0.0f;
#pragma unroll
    for (int s4 = 0; s4 < (S4C > 0 ? S4C : S4); ++s4) {
        const f32x4 a0 = dv[s4 * 64], a1 = dv[s4 * 64 + 32];
        const f32x4 r0 = rw[2 * s4], r1 = rw[2 * s4 + 1];
        acc = fmaf(r0[0], a0[0], acc);
        acc = fmaf(r0[1], a1[0], acc);
        acc = fmaf(r0[2], a0[1], acc);
        acc = fmaf(r0[3], a1[1], acc);
        acc = fmaf(r1[0], a0[2], acc);
        acc = fmaf(r1[1], a1[2], acc);
        acc = fmaf(r1[2], a0[3], acc);
        acc = fmaf(r1[3], a1[3], acc);
    }
    return acc;
}

// Residual sample at global index g as seen by the window of position p.  Inside the signal: the
// sample.  Outside: the initial table is ZERO padded (modeling.py:159-164) but every local update
// REFLECT pads (modeling.py:1046), so a row that has been re-correlated at least once (edge bit set)
// sees r reflected about 0 / T-1, an untouched one sees 0.  (T >= 3W-2 on this path: one bounce.)
//
// One reflected sample can be STALE in the reference's table: with an even W, row T-1 reads r[T-1-W/2]
// through the reflection, an atom at p = T-1-W changes that sample, but only re-correlates rows up to T-2
// (:1028-1049).  The reference keeps row T-1 as it was; so does the score kept here, and the resolution
// of (k, c) must see the old sample too: edge[2] = its index + 1 (0: nothing stale), edge[3] = its bits,
// recorded by the update and dropped when row T-1 is re-correlated.  (No other row / sample / side can
// go stale: for row t < T-1, for odd W and at the left edge every update that changes a reflected sample
// also re-correlates the rows that read it.)
__device__ __forceinline__ float edge_bits_to(unsigned long long b, float) { return __uint_as_float((unsigned)b); }
__device__ __forceinline__ double edge_bits_to(unsigned long long b, double) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ unsigned long long edge_bits_of(float v) { return (unsigned long long)__float_as_uint(v); }
__device__ __forceinline__ unsigned long long edge_bits_of(double v) { return (unsigned long long)__double_as_longlong(v); }

template <typename R>
__device__ __forceinline__ R edge_window_value(const R* __restrict__ r, int T, int g, int p, const unsigned long long* edge)
{
    if (g >= 0 && g < T) return r[g];
    if (g < 0) return ((edge[0] >> p) & 1ull) ? r[-g] : (R)0;
    if (!((edge[1] >> (T - 1 - p)) & 1ull)) return (R)0;
    const int m = 2 * (T - 1) - g;
    if (p == T - 1 && edge[2] == (unsigned long long)(m + 1)) return edge_bits_to(edge[3], (R)0);
    return r[m];
}

__device__ __forceinline__ unsigned long long bit_range(int a, int b)     // bits a..b (0 <= a <= b <= 63)
{
    const unsigned long long hi = b >= 63 ? ~0ull : ((1ull << (b + 1)) - 1ull);
    return hi & ~((1ull << a) - 1ull);
}

// ================================================================================================
// float64 tile: v_mfma_f64_16x16x4_f64 -- [16 atoms] x [16 positions], 4 taps per MFMA.  Measured
// (tools/mfma_f64_probe.hip): bit-exact k-ordered fma chain like the f32 form, 57-70 TFLOP/s.
//   A[i][kk] = D[atom0+i][4s+kk]  (lane l: i = l&15, kk = l>>4)     B[kk][j] = win[pos0+j + 4s+kk]
//   C/D: 4 doubles per lane, col = l&15 (position), row = (l>>4) + 4*reg (atom)
// Image Dimg64[g][c][lane][2]: group g of 16 atoms, chunk c of 8 taps = two k-steps per 16-byte word.
// ================================================================================================
inline void mfma_build_dict_image_f64(const double* D, int K, int W, std::vector<double>& out)
{
    const int G = (K + 15) / 16, S4 = mfma_chunks(W);
    out.assign((size_t)G * S4 * 64 * 2, 0.0);
    for (int g = 0; g < G; ++g)
        for (int c = 0; c < S4; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 2; ++e) {
                    const int k = 16 * g + (lane & 15);
                    const int w = 8 * c + 4 * e + (lane >> 4);
                    if (k < K && w < W) out[(((size_t)g * S4 + c) * 64 + lane) * 2 + e] = D[(size_t)k * W + w];
                }
}

__device__ __forceinline__ int dimg_index_f64(int k, int w, int S4)
{
    const int r8 = w & 7;
    return ((((k >> 4) * S4 + (w >> 3)) * 64 + (k & 15) + 16 * (r8 & 3)) * 2) + (r8 >> 2);
}

template <int S4C>
__device__ __forceinline__ double resolve_chain_f64(const double* __restrict__ dimg, const double* __restrict__ rwin, int k, int S4rt)
{
    const int S4 = S4C > 0 ? S4C : S4rt;
    const f64x2* dv = reinterpret_cast<const f64x2*>(dimg) + ((k >> 4) * S4) * 64 + (k & 15);
    const f64x2* rw = reinterpret_cast<const f64x2*>(rwin);
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < (S4C > 0 ? S4C : S4); ++c) {
        const f64x2 q0 = dv[c * 64], q1 = dv[c * 64 + 16], q2 = dv[c * 64 + 32], q3 = dv[c * 64 + 48];
        const f64x2 r0 = rw[4 * c], r1 = rw[4 * c + 1], r2 = rw[4 * c + 2], r3 = rw[4 * c + 3];
        acc = fma(r0[0], q0[0], acc);      // taps 8c+0 .. 8c+3: k-step 2c, kk = 0..3
        acc = fma(r0[1], q1[0], acc);
        acc = fma(r1[0], q2[0], acc);
        acc = fma(r1[1], q3[0], acc);
        acc = fma(r2[0], q0[1], acc);      // taps 8c+4 .. 8c+7: k-step 2c+1
        acc = fma(r2[1], q1[1], acc);
        acc = fma(r3[0], q2[1], acc);
        acc = fma(r3[1], q3[1], acc);
    }
    return acc;
}

template <int S4C, bool HAS_W>
__device__ __forceinline__ double mfma_tile_score_f64(const double* __restrict__ dimg, const double* __restrict__ win,
                                                      const double* __restrict__ wts, int G, int S4rt, int lane, int& grp)
{
    const int j = lane & 15, kk = lane >> 4;
    const double* wb = win + j + kk;
    double bs = 0.0;
    int bg = 0;                                // group hint, see mfma_tile_score (16-atom groups here)
    const f64x2* dv = reinterpret_cast<const f64x2*>(dimg) + lane;
    auto reduce_elem = [&](double v, int k) {
        if (HAS_W) v = v * wts[k];
        bs = fmax(bs, fabs(v));
    };
    if constexpr (S4C > 0) {
        constexpr int NM = 2 * S4C;            // MFMAs per atom group
        double bop[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) bop[m] = wb[4 * m];
        f64x2 a0[S4C], a1[S4C];
        f64x4 acc0, acc1;
        auto load_a = [&](f64x2 (&a)[S4C], int g) {
#pragma unroll
            for (int c = 0; c < S4C; ++c) a[c] = dv[(g * S4C + c) * 64];
        };
        auto run_first = [&](const f64x2 (&a)[S4C], f64x4& acc) {
            acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m >> 1][m & 1], bop[m], acc, 0, 0, 0);
        };
        auto run_next = [&](const f64x2 (&a)[S4C], f64x4& acc, const f64x4& accp, int gp) {
            const int kbasep = 16 * gp + kk;
            const double before = bs;
            acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m >> 1][m & 1], bop[m], acc, 0, 0, 0);
#pragma unroll
                for (int e = (m * 4) / NM; e < ((m + 1) * 4) / NM; ++e) reduce_elem(accp[e], kbasep + 4 * e);
            }
            bg = bs > before ? gp : bg;
        };
        auto reduce_all = [&](const f64x4& acc, int gl) {
            const double before = bs;
#pragma unroll
            for (int e = 0; e < 4; ++e) reduce_elem(acc[e], 16 * gl + kk + 4 * e);
            bg = bs > before ? gl : bg;
        };
        load_a(a0, 0);
        if (G > 1) load_a(a1, 1);
        run_first(a0, acc0);
        int g = 1;
        for (; g + 1 < G; g += 2) {
            load_a(a0, g + 1);
            run_next(a1, acc1, acc0, g - 1);
            if (g + 2 < G) load_a(a1, g + 2);
            run_next(a0, acc0, acc1, g);
        }
        if (g < G) {
            run_next(a1, acc1, acc0, g - 1);
            reduce_all(acc1, g);
        } else {
            reduce_all(acc0, G - 1);
        }
    } else {
        const int S4 = S4rt;
        for (int g = 0; g < G; ++g) {
            f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < S4; ++c) {
                const f64x2 a = dv[(g * S4 + c) * 64];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], wb[8 * c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], wb[8 * c + 4], acc, 0, 0, 0);
            }
            const double before = bs;
#pragma unroll
            for (int e = 0; e < 4; ++e) reduce_elem(acc[e], 16 * g + kk + 4 * e);
            bg = bs > before ? g : bg;
        }
    }
    // the four 16-lane quarters hold the same position with interleaved atom sets: larger score, on a tie the lower group
#pragma unroll
    for (int x = 16; x <= 32; x <<= 1) {
        const double os = __shfl_xor(bs, x);
        const int og = __shfl_xor(bg, x);
        bg = (os > bs || (os == bs && og < bg)) ? og : bg;
        bs = fmax(bs, os);
    }
    grp = bg;
    return bs;
}

// ---- tile traits: what the kernels below need to know about the two MFMA shapes ----------------
struct TileF32 {
    using R = float;
    static constexpr int TP = 32;                       // positions per tile
    static constexpr int GA = 32;                       // atoms per group
    static constexpr int kChunkElems = 256;             // image elements per (group, chunk): 64 lanes x 16 bytes
    static constexpr size_t kMaxImageBytes = 64 * 1024; // two workgroups per CU
    static int groups(int K) { return (K + 31) / 32; }
    template <int S4C, bool HAS_W>
    static __device__ __forceinline__ R tile_score(const R* dimg, const R* win, const R* wts, int G, int S4, int lane, int& grp)
    { return mfma_tile_score<S4C, HAS_W>(dimg, win, wts, G, S4, lane, grp); }
    template <int S4C, bool HAS_W>
    static __device__ __forceinline__ R tile_score_lean(const R* dimg, const R* win, const R* wts, int G, int, int lane, int& grp)
    { return mfma_tile_score_lean<S4C, HAS_W>(dimg, win, wts, G, lane, grp); }
    template <int S4C> static __device__ __forceinline__ R resolve(const R* dimg, const R* rwin, int k, int S4)
    { return resolve_chain<S4C>(dimg, rwin, k, S4); }
    static __device__ __forceinline__ int dindex(int k, int w, int S4) { return dimg_index(k, w, S4); }
};
struct TileF64 {
    using R = double;
    static constexpr int TP = 16;
    static constexpr int GA = 16;
    static constexpr int kChunkElems = 128;
    static constexpr size_t kMaxImageBytes = 128 * 1024; // one workgroup per CU
    static int groups(int K) { return (K + 15) / 16; }
    template <int S4C, bool HAS_W>
    static __device__ __forceinline__ R tile_score(const R* dimg, const R* win, const R* wts, int G, int S4, int lane, int& grp)
    { return mfma_tile_score_f64<S4C, HAS_W>(dimg, win, wts, G, S4, lane, grp); }
    template <int S4C, bool HAS_W>
    static __device__ __forceinline__ R tile_score_lean(const R* dimg, const R* win, const R* wts, int G, int S4, int lane, int& grp)
    { return mfma_tile_score_f64<S4C, HAS_W>(dimg, win, wts, G, S4, lane, grp); }      // (no four-signal form for float64)
    template <int S4C> static __device__ __forceinline__ R resolve(const R* dimg, const R* rwin, int k, int S4)
    { return resolve_chain_f64<S4C>(dimg, rwin, k, S4); }
    static __device__ __forceinline__ int dindex(int k, int w, int S4) { return dimg_index_f64(k, w, S4); }
};
template <typename R> struct TileOf;
template <> struct TileOf<float> { using type = TileF32; };
template <> struct TileOf<double> { using type = TileF64; };

template <typename R> inline bool mfma_supported(int K, int W, int F)
{
    using Tile = typename TileOf<R>::type;
    if (F != 1) return false;   // (per launch, T >= 3W-2 is required as well)
    const size_t bytes = (size_t)Tile::groups(K) * mfma_chunks(W) * Tile::kChunkElems * sizeof(R);
    return bytes <= Tile::kMaxImageBytes && W <= 128;
}

// GS = signals per workgroup.  1: the signal owns a 256-thread workgroup (two per CU: each holds its own 64 KB image).
// 4: four signals share a 1024-thread workgroup and ONE image (float32 only): four waves per SIMD, each from a
// different signal, so that while one signal is in its serial phases (loads, resolve, residual update, maxima, stop
// rules) the matrix pipe has three other signals' tiles to run.  The waves of a signal meet at SoftSync barriers.
template <typename Tile, int S4C, bool HAS_W, int GS = 1> struct MfmaRecorr {
    static constexpr int kMaxSegments = kMfmaMaxSeg;
    static constexpr bool kFused = true;
    static constexpr bool kLocomp = false;
    static constexpr int kGroup = GS;
    // -DHSCMP_QUAD_LOCKSTEP=1 (measurement only): barriers B1 and B4 of the atom body become hardware barriers across
    // the four signals of the workgroup, which lines their tiles up -- all serial phases then run together without a
    // matrix instruction beside them, all tiles together.  Measured 10.1-10.2 ms against 9.6 ms for the greedy loop of
    // config 2: what the serial phases gain by running alone (tools/serial_stretch_probe.hip) is less than what the
    // matrix pipe loses by idling through them.
#ifndef HSCMP_QUAD_LOCKSTEP
#define HSCMP_QUAD_LOCKSTEP 0
#endif
    static constexpr bool kLockstep = GS > 1 && HSCMP_QUAD_LOCKSTEP != 0;
    static constexpr int kMinWavesPerSimd = GS;         // (launch bounds: 4 signals x 4 waves = 4 waves per SIMD)
    using Sync = typename std::conditional<GS == 1, HwSync, SoftSync>::type;
    static constexpr int kEnergyWaves = kWaves;
    static constexpr bool kScoreOnly = true;    // best_c[t] holds max_k |c[t,k]*w_k|; (k, c) resolved on selection
    static constexpr int kBook = 192;           // bookkeeping thread: lane 0 of wave 3, idle while waves 0.. rescan segments
    using R = typename Tile::R;
    static constexpr int TP = Tile::TP;
    static __device__ __forceinline__ const R* weights(const DevParams&, const State<R>& S, const MfmaArgsT<R>&, char*) { return S.weights; }
    static __device__ __forceinline__ void on_atom(const DevParams&, const State<R>&, const MfmaArgsT<R>&, char*, int, int) {}
    static __device__ __forceinline__ bool update_residual(const DevParams&, const State<R>&, const Sig<R>&, const MfmaArgsT<R>&, char*, int, int, R,
                                                           int, int, int, R&, R&) { return false; }
    static __device__ __forceinline__ bool window_partials(const DevParams&, const Sig<R>&, const MfmaArgsT<R>&, char*, int, int, R&) { return false; }
    static __device__ __forceinline__ bool wave_window_listed(const DevParams&, const Sig<R>&, const MfmaArgsT<R>&, char*, int, int, int, int, R&) { return false; }
    static __device__ __forceinline__ bool row_results(const DevParams&, const MfmaArgsT<R>&, char*, int, const int*&, const R*&, const R*&, int&, int&) { return false; }
    static __device__ __forceinline__ bool residual_copy_in_lds(const MfmaArgsT<R>&, char*) { return false; }
    using Shared = IterSharedT<R, kMfmaMaxSeg, false, false>;
    using Args = MfmaArgsT<R>;
    static __device__ __forceinline__ Sync make_sync(Shared& sh)
    {
        if constexpr (GS == 1) return HwSync();
        else { SoftSync sy; sy.init(&sh.bar, &sh.bar_cnt); return sy; }
    }

    struct Layout {
        R* dimg; R* wts; R* win; R* esq; R* sbs; unsigned* bloom;
        R* rwin; R* rwin_w; unsigned long long* edge;
        int nwin, wp, nsbmax;
    };

    static __host__ __device__ int window_floats(int W, int S4) { return ((2 * W - 1 + TP - 1) / TP) * TP + 8 * S4 + 32; }
    static __host__ __device__ int segbuf_len(int W, int seg) { return ((2 * W - 2) / seg + 2) * seg; }
    // (the same with the segment length as a shift: P.seg == 1 << P.seg_shift -- no integer division on the device)
    static __host__ __device__ int segbuf_len_p(const DevParams& P) { return (((2 * P.W - 2) >> P.seg_shift) + 2) << P.seg_shift; }
    // LDS: what the signals of a workgroup share (dictionary image, weights), then per signal the control block and
    // its windows.  GS == 1: [control][image | weights | windows ...] as one region behind the control block.
    static __host__ __device__ size_t shared_lds_bytes(const Args& A)
    {
        return ((size_t)A.G * A.S4 * Tile::kChunkElems + (HAS_W ? Tile::GA * A.G : 0)) * sizeof(R);      // (a multiple of 16)
    }
    static __host__ __device__ size_t private_lds_bytes(const DevParams& P, const Args& A)
    {
        const size_t relems = (size_t)window_floats(P.W, A.S4) + 2 * 8 * A.S4 + (size_t)segbuf_len_p(P) + 8 * A.S4 + kWaves * 8 * A.S4;
        return relems * sizeof(R) + kBloomWords * sizeof(unsigned) + kEdgeWords * sizeof(unsigned long long);
    }
    static __host__ __device__ size_t per_signal_lds_bytes(const DevParams& P, const Args& A)
    {
        return ((sizeof(Shared) + 15) / 16) * 16 + ((private_lds_bytes(P, A) + 15) / 16) * 16;
    }
    static size_t extra_lds_bytes(const DevParams& P, const Args& A) { return shared_lds_bytes(A) + private_lds_bytes(P, A); }   // GS == 1
    static size_t total_lds_bytes(const DevParams& P, const Args& A)
    {
        if (GS == 1) return ((sizeof(Shared) + 15) / 16) * 16 + extra_lds_bytes(P, A);
        return shared_lds_bytes(A) + (size_t)GS * per_signal_lds_bytes(P, A);
    }
    static __device__ __forceinline__ int signal_lds_offset(const DevParams& P, const Args& A)
    {
        // (readfirstlane: the sizes involve an integer division, which the compiler carries out on the vector ALU -- the
        // uniform result would sit in a VGPR, and with it every LDS address derived from it, live across the whole loop)
        if constexpr (GS == 1) return 0;
        else return __builtin_amdgcn_readfirstlane((int)shared_lds_bytes(A) + gsig() * (int)per_signal_lds_bytes(P, A));
    }
    static __device__ __forceinline__ Layout layout(const DevParams& P, const Args& A, char* lds)
    {
        const int S4 = S4C > 0 ? S4C : A.S4;
        Layout L;
        L.dimg = reinterpret_cast<R*>(GS == 1 ? lds : dyn_lds());
        L.wts = L.dimg + A.G * S4 * Tile::kChunkElems;
        L.nwin = window_floats(P.W, S4);
        L.wp = 8 * S4;
        L.nsbmax = segbuf_len_p(P);
        L.win = GS == 1 ? L.wts + (HAS_W ? Tile::GA * A.G : 0) : reinterpret_cast<R*>(lds);
        L.esq = L.win + L.nwin;
        L.sbs = L.esq + 2 * L.wp;
        L.bloom = reinterpret_cast<unsigned*>(L.sbs + L.nsbmax);
        L.rwin = reinterpret_cast<R*>(L.bloom + kBloomWords);
        L.rwin_w = L.rwin + L.wp;
        L.edge = reinterpret_cast<unsigned long long*>(L.rwin_w + kWaves * L.wp);
        return L;
    }

    // what the signals of a workgroup share: image and weights by all its threads, the barrier counters, then the
    // ONE hardware barrier of the kernel (every wave is still here: nobody has returned yet)
    static __device__ __forceinline__ void prologue_shared(const DevParams& P, const State<R>& S, const Args& A, char* smem)
    {
        if constexpr (GS > 1) {
            const int S4 = S4C > 0 ? S4C : A.S4;
            R* dimg = reinterpret_cast<R*>(smem);
            R* wts = dimg + A.G * S4 * Tile::kChunkElems;
            lds_copy16(dimg, A.dimg, A.G * S4 * Tile::kChunkElems * (int)sizeof(R), (int)threadIdx.x, GS * kThreads);
            if (HAS_W) for (int i = threadIdx.x; i < Tile::GA * A.G; i += GS * kThreads) wts[i] = i < P.K ? S.weights[i] : (R)0;
            if (ltid() == 0) {
                Shared* sh = reinterpret_cast<Shared*>(smem + signal_lds_offset(P, A));
                sh->bar = 0u; sh->bar_cnt = 0;
            }
            __syncthreads();
        }
    }

    static __device__ __forceinline__ void prologue(const DevParams& P, const State<R>& S, const Args& A, char* lds, int b, Sync& sy)
    {
        const Layout L = layout(P, A, lds);
        const int S4 = S4C > 0 ? S4C : A.S4;
        const int tid = ltid();
        if constexpr (GS == 1) {
            lds_copy16(L.dimg, A.dimg, A.G * S4 * Tile::kChunkElems * (int)sizeof(R));
            if (HAS_W) for (int i = tid; i < Tile::GA * A.G; i += kThreads) L.wts[i] = i < P.K ? S.weights[i] : (R)0;
        }
        for (int i = tid; i < L.nwin; i += kThreads) L.win[i] = (R)0;   // the tail behind the span stays zero
        for (int i = tid; i < kBloomWords; i += kThreads) L.bloom[i] = 0u;
        for (int i = tid; i < (1 + kWaves) * L.wp; i += kThreads) L.rwin[i] = (R)0;   // padded taps stay zero
        if (tid < kEdgeWords) L.edge[tid] = S.edge[kEdgeWords * b + tid];
        sy.full();
        // resumed launch: re-enter the (t,k) pairs selected so far
        const int nslots = S.stats[(int64_t)b * ST_COUNT + ST_SLOTS];
        const int* st = S.slot_t + (int64_t)b * P.cap;
        const int* sk = S.slot_k + (int64_t)b * P.cap;
        for (int i = tid; i < nslots; i += kThreads) {
            const unsigned h = bloom_hash(st[i], sk[i]);
            atomicOr(&L.bloom[h >> 5], 1u << (h & 31));
        }
        // visibility: the caller's next barrier
    }

    static __device__ __forceinline__ void epilogue(const DevParams& P, const State<R>& S, const Args& A, char* lds, int b)
    {
        const Layout L = layout(P, A, lds);
        if (ltid() < kEdgeWords) S.edge[kEdgeWords * b + ltid()] = L.edge[ltid()];
    }

    // never reached: iterate_kernel hands the whole atom body to apply_atom() when kFused
    template <typename SH>
    static __device__ __forceinline__ void run(const DevParams&, const State<R>&, const Sig<R>&, SH&, const Args&, char*, int, int) {}

    // (k, c) of position t by ONE wave (blocked selection, modeling.py:935-946): the window goes to
    // this wave's private LDS strip, lanes stride over the atoms, first k wins ties.
    static __device__ __forceinline__ void resolve_wave(const DevParams& P, const State<R>&, const Sig<R>& Gs,
                                                        const Args& A, char* lds, int t, int lane, int& k_out, R& c_out)
    {
        const Layout L = layout(P, A, lds);
        const int S4 = S4C > 0 ? S4C : A.S4;
        R* rw = L.rwin_w + (ltid() >> 6) * L.wp;
        __builtin_amdgcn_wave_barrier();
        const int gh = Gs.bk[t];                            // the hint travels with the window's samples
        for (int w = lane; w < P.W; w += 64) rw[w] = edge_window_value(Gs.r, P.T, t - P.off + w, t, L.edge);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        resolve_group(P, L, rw, gh, lane, S4, k_out, c_out);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    // (k, c) of a position from its window `rw` (LDS, this wave's strip): the tile that wrote the position's score also
    // wrote which atom group holds the first atom attaining it (`hint` = the loaded Gs.bk[t], uniform; see
    // mfma_tile_score), so one chain per lane over that group's atoms is all there is to recompute; first k wins ties
    // (:970).  One wave, nobody else's data: result in every lane.
    static __device__ __forceinline__ void resolve_group(const DevParams& P, const Layout& L, const R* rw,
                                                         int hint, int lane, int S4, int& k_out, R& c_out)
    {
        const int g = __builtin_amdgcn_readfirstlane(hint);
        const int k = Tile::GA * g + lane;
        Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
        R bc = (R)0;
        if (lane < Tile::GA && k < P.K) {
            bc = Tile::template resolve<S4C>(L.dimg, rw, k, S4);
            if (HAS_W) { const R sw = bc * L.wts[k]; best.s = rabs(sw); } else best.s = rabs(bc);
            // (a NaN -- diverged pursuit -- must not reach the reduction of the float64 build, whose comparisons would leave
            //  the lanes with different winners; the float32 reduction orders bit patterns, where a NaN is the largest)
            if constexpr (sizeof(R) == 8) { if (best.s != best.s) best.s = (R)INFINITY; }
            best.i = k;
        }
        best = wave_argmax_first(best);                     // (lane l holds atom GA*g + l: lanes in index order)
        k_out = best.i;
        c_out = wave_bcast(bc, best.i & (Tile::GA - 1));
    }

    // One applied atom at position p: modeling.py:1106-1142.  resolved: (k, c) already known (blocked
    // selection); otherwise they are resolved here and the null test of :974 is applied.
    // Returns true when the atom loop must stop.
    template <typename SH>
    static __device__ __forceinline__ bool apply_atom(const DevParams& P, const State<R>& S, const Sig<R>& Gs,
                                                      SH& sh, const Args& A, char* lds, int p, int k, R c, bool resolved, Sync& sy,
                                                      FusedCtl& fc)
    {
        // single arg-max rounds without a residual-scale rule: the round's own bookkeeping (:1160-1163) is done here
        const bool round_is_atom = !P.blocked && !P.has_scale;
        (void)S;
        // p (and k, c once known) are wave-uniform: telling the compiler moves the span / segment arithmetic that
        // derives from them to the scalar unit (the vector ALU is what the f32 MFMA competes for)
        p = __builtin_amdgcn_readfirstlane(p);
        // Four signals per workgroup (128 VGPRs): the thread index is laundered once per atom.  Everything derived from it
        // (LDS addresses of the windows, segment buffer, squares ...) is loop invariant; hoisted out of the atom loop it
        // is live across the MFMA tile, gets spilled, and every reload is a scratch round trip on the atom's critical
        // path.  Recomputing a few shifts and adds per atom is cheaper.
        int tid = ltid();
        if constexpr (GS > 1) asm volatile("" : "+v"(tid));
        const int T = P.T, W = P.W, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int S4 = S4C > 0 ? S4C : A.S4;
        const Layout L = layout(P, A, lds);
        if (fc.nev >= P.cap) {                                  // event list full
            if (tid == 0) { sh.converged = 1; sh.stop = STOP_CAPACITY; if (round_is_atom) { sh.rounds += 1; sh.offset = !sh.offset; } }
            sy.full();
            return true;
        }
        HSCMP_STAMP_BEGIN();
        const int nrows = 2 * W - 1, ntiles = (nrows + TP - 1) / TP, span = 3 * W - 2;
        const int tstart = p - P.off - (W - 1);                 // :1028-1033
        const int tend = p + W / 2 + (W - 1);                   // :1038
        const int sidx = tstart < 0 ? 0 : tstart;               // :1034
        const int eidx = tend > T - 1 ? T - 1 : tend;           // :1039
        const int nslice = eidx - sidx + 1;
        const bool interior = tstart >= 0 && tend <= T - 1;
        int s, e, es;
        const int len = centered_span(T, W, p, s, e, es);       // support of the atom (utils.py:103-131)
        const int lo = max(0, p - (W - 1)), hi = min(T - 1, p + (W - 1));
        const int sg0 = lo >> P.seg_shift, sg1 = hi >> P.seg_shift;
        const int segbase = (sg0 << P.seg_shift);
        const int nsb = min(T, ((sg1 + 1) << P.seg_shift)) - segbase;    // positions of the touched segments

        if (resolved) { k = __builtin_amdgcn_readfirstlane(k); c = wave_bcast(c, 0); }
        HSCMP_MARK("A_loads");
        // ---- phase A: every global load of this atom, issued together -------------------------
        // first what the longest chain of the atom waits for: the position's window and its group hint.  EVERY wave
        // loads them and resolves (k, c) for itself (64 samples and one word, four times): no barrier, no exchange
        // Compile-time chunk count: W <= 8 * S4C bounds the span (3W - 2) and the window (W), so a kernel built for
        // W <= 64 has no second pass over either.  And away from the signal ends -- a uniform branch, almost every
        // atom -- there is no reflection (an integer modulo on the vector ALU) and no edge history to consult.
        constexpr int kUS = (S4C > 0 && 24 * S4C - 2 <= kThreads) ? 1 : 2;      // passes over the span
        constexpr int kUW = (S4C > 0 && 8 * S4C <= 64) ? 1 : 2;                 // passes over the window
        int gh = 0;
        R wres[2] = {(R)0, (R)0};                               // W <= 128: two taps per lane
        R rv[2] = {(R)0, (R)0}; int rm[2] = {-1, -1};
        if (interior) {
            if (!resolved) {
                gh = Gs.bk[p];
#pragma unroll
                for (int u = 0; u < kUW; ++u) if (lane + 64 * u < W) wres[u] = Gs.r[p - P.off + lane + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < kUS; ++u) {
                const int i = tid + u * kThreads;
                if (i < span) { rm[u] = tstart + i; rv[u] = Gs.r[rm[u]]; }
            }
        } else {
            if (!resolved) {
                gh = Gs.bk[p];
#pragma unroll
                for (int u = 0; u < kUW; ++u) if (lane + 64 * u < W) wres[u] = edge_window_value(Gs.r, T, p - P.off + lane + 64 * u, p, L.edge);
            }
#pragma unroll
            for (int u = 0; u < kUS; ++u) {
                const int i = tid + u * kThreads;
                if (i < span) {                                 // np.pad 'reflect', :1046
                    rm[u] = reflect_index(tstart + i, sidx, nslice);
                    rv[u] = Gs.r[rm[u]];
                }
            }
        }
        R os[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kThreads;
            os[u] = (R)0;
            if (i < nsb) os[u] = Gs.bc[segbase + i];            // old scores of the touched segments
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kThreads;
            if (i < nsb) L.sbs[i] = os[u];
        }
        // signals longer than 131072 samples have segments of 512+ positions: the rest of the touched segments
        for (int i = tid + 2 * kThreads; i < nsb; i += kThreads) L.sbs[i] = Gs.bc[segbase + i];

        HSCMP_STAMP(8);                                         // phase A issued
        HSCMP_MARK("resolve");
        // ---- resolve (k, c) of the selected position (:970) ------------------------------------
        if (!resolved) {
            R* rw = L.rwin_w + wv * L.wp;                   // this wave's strip
#pragma unroll
            for (int u = 0; u < kUW; ++u) if (lane + 64 * u < W) rw[lane + 64 * u] = wres[u];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            resolve_group(P, L, rw, gh, lane, S4, k, c);
            k = __builtin_amdgcn_readfirstlane(k);
            c = wave_bcast(c, 0);
            if (P.has_thres && !(fabs((double)c) > P.thres)) {  // :974 null coefficient: empty selection
                // (with a residual-scale rule the reference tests that rule first, :1145-1153: the slow rules name the reason)
                if (tid == 0) {
                    sh.converged = 1; sh.nullsel = 1; if (!P.has_scale && sh.stop == STOP_RUNNING) sh.stop = STOP_EMPTY;
                    if (round_is_atom) { sh.rounds += 1; sh.offset = !sh.offset; }
                }
                sy.full();
                return true;
            }
        }

        HSCMP_STAMP(9);                                         // resolve
        HSCMP_MARK("dupcheck");
        // duplicate check (:1106): Bloom filter in LDS; only a hit pays for the scan of the slot list
        // Long slot lists (sh.hashed, see slot_find): the bookkeeping thread probes the hash table instead; its
        // first probe is in flight under the atom's work and is only looked at in the bookkeeping below.
        const bool hashed = fc.hashed != 0;                     // uniform: fetched behind the previous atom's last barrier
        const unsigned hb = bloom_hash(p, k);
        const bool maybe_dup = !hashed && ((L.bloom[hb >> 5] >> (hb & 31)) & 1u) != 0;      // uniform
        unsigned long long probe_key = kSlotEmpty;
        unsigned probe_pos = 0;
        if (maybe_dup) {
            sy.full();                                    // drains the bookkeeper's deferred slot stores
            const int nslots = sh.nslots;
            for (int i = tid; i < nslots; i += kThreads)
                if (Gs.slot_t[i] == p && Gs.slot_k[i] == k) sh.found = i;      // at most one match
        } else if (hashed && tid == kBook) {
            probe_pos = slot_hash(slot_key(p, k)) & P.hmask;
            probe_key = hkey_load(Gs.hkey + probe_pos);
        }

        HSCMP_MARK("residual");
        // ---- residual subtract (:1117, :996-1016) on the register copy; window + squares to LDS
        const R nc = -c;
        bool own[2] = {false, false};                           // this thread owns the sample itself: it stores the new value
        if (interior) {                                         // no reflected copies, no edge history
#pragma unroll
            for (int u = 0; u < kUS; ++u) {
                const int i = tid + u * kThreads;
                if (i < span) {
                    R v = rv[u];
                    const int q = rm[u] - s;
                    if (q >= 0 && q < e - s) {
                        const R prod = nc * L.dimg[Tile::dindex(k, es + q, S4)];   // -c*D[k] rounded, then += (utils.py:120,129)
                        const R vn = v + prod;
                        // (W = 2 only: the atom at T-1-W is an interior one, see the stale-sample note below)
                        if (!(W & 1) && p == T - 1 - W && rm[u] == T - 1 - W / 2 && (L.edge[1] & 1ull) && L.edge[2] == 0ull) {
                            L.edge[3] = edge_bits_of(v);
                            L.edge[2] = (unsigned long long)(rm[u] + 1);
                        }
                        own[u] = true;
                        L.esq[q] = v * v;
                        L.esq[L.wp + q] = vn * vn;
                        v = vn;
                    }
                    L.win[i] = v;
                    rv[u] = v;
                }
            }
        } else
#pragma unroll
        for (int u = 0; u < kUS; ++u) {
            const int i = tid + u * kThreads;
            if (i < span) {
                R v = rv[u];
                const int m = rm[u];
                if (m >= s && m < e) {
                    const int q = m - s;
                    const R prod = nc * L.dimg[Tile::dindex(k, es + q, S4)];   // -c*D[k] rounded, then += (utils.py:120,129)
                    const R vn = v + prod;
                    if (tstart + i == m) {                    // the sample itself (not a reflected copy)
                        // even W: an atom at T-1-W changes the sample that row T-1 reads through the reflection
                        // without re-correlating that row -- keep what row T-1 saw (see edge_window_value)
                        if (!(W & 1) && p == T - 1 - W && m == T - 1 - W / 2 && (L.edge[1] & 1ull) && L.edge[2] == 0ull) {
                            L.edge[3] = edge_bits_of(v);
                            L.edge[2] = (unsigned long long)(m + 1);
                        }
                        own[u] = true;
                        L.esq[q] = v * v;
                        L.esq[L.wp + q] = vn * vn;
                    }
                    v = vn;
                }
                L.win[i] = v;
                rv[u] = v;
            }
        }
        HSCMP_STAMP(0);                                         // phase A + resolve + update, up to B1
        // B1: window, squares, segment buffer in LDS.  (Lockstep build: waves of a signal that has finished have ended
        // and the hardware barrier does not count them; every wave passes exactly two hardware barriers per atom, B1
        // and B4, and leaves the loop only between atoms, so all signals are always at the same one of the two.)
        if constexpr (kLockstep) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        else sy.lds();
        // The new samples go to memory only BEHIND this barrier.  Near a signal end a sample of the atom's support is
        // loaded twice in phase A -- by its own thread and, through the reflection, by a thread of another wave -- and
        // with blocked selection nothing else separates those loads from this store: a wave that left the previous
        // barrier late would load the sample after it had changed and subtract the atom from it a second time.  Every
        // thread has consumed what it loaded before it arrives here (its window entry depends on it).
#pragma unroll
        for (int u = 0; u < kUS; ++u) if (own[u]) Gs.r[rm[u]] = rv[u];
        if (P.has_scale) {                                      // toleranceResidualScale: max|r| of touched segments
            sy.full();                                          // (the stores above are visible to the scan)
            for (int sg = (s >> P.seg_shift) + wv; sg <= ((e - 1) >> P.seg_shift); sg += kWaves) rscan_segment(P, Gs, sh, sg, lane);
        }
        HSCMP_STAMP(1);                                         // B1
        HSCMP_MARK("energy");
        // local energy before / after (:1002-1005): pinned tree, partial q lives in thread q
        auto energy_partials = [&]() {
            R pb = (R)0, pa = (R)0;
            if (wv * 64 < len) {                                // (a wave without samples: its tree of zeros sums to +0)
                if (tid < len) { pb = L.esq[tid]; pa = L.esq[L.wp + tid]; }
                wave_tree_down2(pb, pa);
            }
            if (lane == 0) { sh.red[wv] = pb; sh.red[kWaves + wv] = pa; }
        };
        if constexpr (!kLockstep) energy_partials();
        HSCMP_STAMP(2);                                         // energy partials

        HSCMP_MARK("tile");
        // ---- local re-correlation of the 2W-1 touched rows on the matrix cores (:1120, :1018-1051)
        // Four waves per SIMD: serial code runs at priority 3, the tiles at priority 0.  Against waves that issue MFMAs
        // back to back a raised priority buys nothing (tools/serial_stretch_probe.hip), but the tiles of this kernel wait
        // for their LDS operands often enough that it is worth 2.5 % here (loop of config 2: 9.56 ms with, 9.84 without).
        if constexpr (GS > 1 && !kLockstep) __builtin_amdgcn_s_setprio(0);
        for (int q = wv; q < ntiles; q += kWaves) {
            R sc;
            int grp;
            if constexpr (GS > 1 && S4C > 0) sc = Tile::template tile_score_lean<S4C, HAS_W>(L.dimg, L.win + TP * q, L.wts, A.G, S4, lane, grp);
            else sc = Tile::template tile_score<S4C, HAS_W>(L.dimg, L.win + TP * q, L.wts, A.G, S4, lane, grp);
            const int row = TP * q + lane, t = p - (W - 1) + row;
            if (lane < TP && row < nrows && t >= 0 && t < T) {  // overlapReplace clipping (utils.py:133-161)
                Gs.bc[t] = sc;
                Gs.bk[t] = grp;                                 // the group hint of the row (resolve_group)
                L.sbs[t - segbase] = sc;
            }
        }
        if constexpr (GS > 1 && !kLockstep) __builtin_amdgcn_s_setprio(3);
        HSCMP_STAMP(3);                                         // MFMA tile(s) of this wave
        // B4: per-row scores in the segment buffer
        if constexpr (kLockstep) { energy_partials(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        else sy.lds();
        HSCMP_STAMP(4);                                         // B4

        HSCMP_MARK("segmax");
        // ---- maxima of the touched segments, out of LDS
        for (int sg = sg0 + wv; sg <= sg1; sg += kWaves) {
            const int t0 = (sg << P.seg_shift), t1 = min(T, t0 + P.seg);
            Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
            const int per = P.seg >> 6;                         // consecutive positions per lane (wave_argmax_first)
            for (int j = 0; j < per; ++j) {
                const int t = t0 + lane * per + j;
                if (t < t1) {
                    const R sc = L.sbs[t - segbase];
                    if (sc > best.s) { best.s = sc; best.i = t; }
                }
            }
            best = wave_argmax_first(best);
            if (lane == 0) { sh.seg_score[sg] = best.s; sh.seg_t[sg] = best.i; }
        }
        HSCMP_MARK("bookkeeping");
        // ---- bookkeeping and the fast stop rules (:1106-1142) on the bookkeeping thread
        int si = -1, ev = 0;
        bool new_slot = false;
        double acc_old = 0.0;
        if (tid == kBook) {
            const R b01 = sh.red[0] + sh.red[1], b23 = sh.red[2] + sh.red[3];
            const R a01 = sh.red[kWaves + 0] + sh.red[kWaves + 1], a23 = sh.red[kWaves + 2] + sh.red[kWaves + 3];
            const R e_before = b01 + b23, e_after = a01 + a23;
            const R loss = e_before - e_after;              // :1005
            sh.e_res = sh.e_res - loss;                         // :1014
            if (hashed) {
                const unsigned long long key = slot_key(p, k);
                for (;;) {                                      // the rest of the linear probe (usually none)
                    if (probe_key == key) { si = hval_load(Gs.hval + probe_pos); break; }
                    if (probe_key == kSlotEmpty) break;
                    probe_pos = (probe_pos + 1) & P.hmask;
                    probe_key = hkey_load(Gs.hkey + probe_pos);
                }
            } else {
                si = sh.found;
                sh.found = -1;
            }
            if (si >= 0) { acc_old = Gs.slot_a[si]; }           // rare: re-selection of an existing (t,k)
            if (si >= 0 && fabs(acc_old) > 0.0) sh.ndup += 1;
            else if (rabs(c) > (R)0) sh.nnz += 1;
            if (si < 0) { new_slot = true; si = sh.nslots++; if (!hashed) L.bloom[hb >> 5] |= 1u << (hb & 31); }
            ev = sh.nev++;
            sh.iters += 1;
            // rows lo..hi now carry reflect-padded values (see edge_window_value)
            if (lo < P.off) L.edge[0] |= bit_range(lo, min(hi, P.off - 1));
            const int rt0 = T - (W - 1 - P.off);                // first position whose window passes T-1
            if (hi >= rt0) L.edge[1] |= bit_range(T - 1 - hi, T - 1 - max(lo, rt0));
            if (hi >= T - 1) L.edge[2] = 0ull;                  // row T-1 re-correlated: nothing stale any more
            if ((double)sh.e_res < P.eps) { sh.converged = 1; sh.stop = STOP_ENERGY_EPS; }
            else if (P.l0 >= 0 && sh.nnz >= P.l0) { sh.converged = 1; sh.stop = STOP_NNZ; }
            else if (P.has_snr) {
                const R qv = sh.e_sig / sh.e_res;
                if ((double)qv >= P.snr_ratio) { sh.converged = 1; sh.stop = STOP_SNR; }
            }
            if (round_is_atom) { sh.rounds += 1; sh.offset = !sh.offset; }
            // what the other waves need to know about this atom, in one 16-byte word (FusedCtl)
            *reinterpret_cast<int4*>(sh.ctl) = make_int4(sh.converged, sh.nev, sh.nslots, hashed ? 1 : 0);
        }
        HSCMP_MARK("B5");
        HSCMP_STAMP(5);                                         // segment maxima + bookkeeping
        sy.full();                                        // B5: also drains this atom's residual / score stores
        HSCMP_STAMP(6);                                         // B5
        if (tid == kBook) {
            // deferred global stores of the bookkeeping: nobody waits for them (the duplicate scan
            // re-synchronises before it reads the slot list), they retire under the next atom's MFMAs
            if (new_slot) {
                Gs.slot_t[si] = p; Gs.slot_k[si] = k;
                if (hashed) slot_insert_at(Gs, probe_pos, p, k, si);            // the probe's free entry
            }
            Gs.slot_a[si] = acc_old + (double)c;                // :1114, :992  (0.0 + c for a new slot)
            Gs.ev_t[ev] = p; Gs.ev_k[ev] = k; Gs.ev_c[ev] = c;
        }
        HSCMP_STAMP(7);                                         // deferred stores
#ifdef HSCMP_DBG_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[15] += 1;
#endif
#ifdef HSCMP_DBG_CHECKSEG
        // diagnostic build: the re-correlated rows near the signal ends again, one thread per row, pinned chain from the (now final) residual
        if (lo < 64 || hi >= T - 64) {
            sy.full();
            const int row = tid, t = p - (W - 1) + row;
            if (row < nrows && t >= 0 && t < T) {
                float best = 0.0f;
                for (int kk = 0; kk < P.K; ++kk) {
                    float acc = 0.0f;
                    for (int w = 0; w < W; ++w)
                        acc = fmaf((float)edge_window_value(Gs.r, T, t - P.off + w, t, L.edge), (float)L.dimg[Tile::dindex(kk, w, S4)], acc);
                    if (HAS_W) acc = acc * (float)L.wts[kk];
                    best = fmaxf(best, fabsf(acc));
                }
                const float kept = (float)Gs.bc[t];
                if (best != kept) {
                    const unsigned long long n = atomicAdd(&g_cnt[1], 1ull);
                    if (n < 3) {
                        g_cnt[4 + 4 * n + 0] = ((unsigned long long)(blockIdx.x * GS + gsig()) << 32) | (unsigned)t | 0x80000000u;
                        g_cnt[4 + 4 * n + 1] = ((unsigned long long)__float_as_uint(best) << 32) | __float_as_uint(kept);
                        g_cnt[4 + 4 * n + 2] = ((unsigned long long)(unsigned)p << 32) | (unsigned)sh.iters;
                        // first tap whose window sample differs from what the tile saw
                        int wbad = -1; float we = 0.f, wg = 0.f;
                        for (int w = 0; w < W && wbad < 0; ++w) {
                            const float ex = (float)edge_window_value(Gs.r, T, t - P.off + w, t, L.edge);
                            const float got = (float)L.win[row + w];
                            if (ex != got) { wbad = w; we = ex; wg = got; }
                        }
                        int mfound = 0xffff;
                        for (int mm = sidx; mm <= eidx; ++mm) if ((float)Gs.r[mm] == wg) { mfound = mm - sidx; break; }
                        (void)we;
                        g_cnt[4 + 4 * n + 3] = ((unsigned long long)(unsigned)(wbad & 0xffff) << 48) | ((unsigned long long)(unsigned)(mfound & 0xffff) << 32) | __float_as_uint(wg);
                        g_cnt[4 + 4 * n + 2] = ((unsigned long long)(unsigned)p << 32) | (unsigned)(sidx & 0xffff) << 16 | (unsigned)(row & 0xffff);
                    }
                }
            }
            sy.full();
        }
#endif
        HSCMP_MARK("atom_end");
        fc = fused_ctl_fetch(sh);
        return fc.converged != 0;
    }
};

// host-side dispatch -----------------------------------------------------------------------------
// CUs of the current device (queried per device: a process may drive several GPUs)
inline int mfma_device_cus()
{
    static int cus_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus_of[dev] == 0) {
        hipDeviceProp_t prop;
        cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return cus_of[dev];
}

template <typename Tile, int S4C, bool HAS_W>
static int mfma_launch_corr_init_t(hipStream_t stream, const DevParams& P, const State<typename Tile::R>& S,
                                   const MfmaArgsT<typename Tile::R>& A, bool dry = false)
{
    using R = typename Tile::R;
    const size_t lds = ((size_t)A.G * A.S4 * Tile::kChunkElems + Tile::GA * A.G + kMfmaChunk + 8 * A.S4 + 32) * sizeof(R);
    auto kern = corr_init_mfma_kernel<Tile, S4C, HAS_W>;
    if (set_dyn_lds((const void*)kern, lds) != hipSuccess) return -1;
    if (dry) return 0;
    // persistent grid: as many workgroups as are resident at once (LDS-bound), capped by the work
    const int cus = mfma_device_cus();
    const int per_cu = cached_blocks_per_cu((const void*)kern, kThreads, lds);
    const int64_t nitems = (int64_t)((P.T + kMfmaChunk - 1) / kMfmaChunk) * P.B;
    int64_t grid = (int64_t)cus * per_cu;
    if (const char* e = getenv("HSCMP_INIT_PER_CU")) grid = (int64_t)cus * std::max(1, atoi(e));      // diagnostic: fewer resident workgroups
    if (grid > nitems) grid = nitems;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kThreads), lds, stream, P, S, A);
    return 0;
}

// signals per workgroup of the last MFMA loop this thread launched (reported by hscmp_last_variant)
inline int& mfma_last_group() { static thread_local int g = 1; return g; }

template <typename Tile, int S4C, bool HAS_W, int GS>
static int mfma_launch_iterate_g(hipStream_t stream, const DevParams& P0, const State<typename Tile::R>& S,
                                 const MfmaArgsT<typename Tile::R>& A, bool dry)
{
    using Pol = MfmaRecorr<Tile, S4C, HAS_W, GS>;
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    size_t lds = Pol::total_lds_bytes(P, A);
    if (GS > 1 && lds > (size_t)160 * 1024) return -1;
    if (const char* pad = getenv("HSCMP_LDS_PAD")) lds += (size_t)atoi(pad);      // diagnostic: force a lower occupancy
    auto kern = iterate_kernel<typename Tile::R, Pol>;
    if (set_dyn_lds((const void*)kern, lds) != hipSuccess) return -1;
    if (dry) return 0;
    if (getenv("HSCMP_DEBUG")) {
        int per_cu = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, GS * kThreads, lds);
        fprintf(stderr, "[hscmp] iterate_kernel<mfma %s S4=%d w=%d, %d signal(s) per workgroup>: dynamic LDS %zu B (control %zu B), occupancy API %d blocks/CU (%s), seg=%d nseg=%d\n",
                sizeof(typename Tile::R) == 4 ? "f32" : "f64", S4C, (int)HAS_W, GS, lds, sizeof(typename Pol::Shared), per_cu,
                hipGetErrorString(e), P.seg, P.nseg);
    }
    hipLaunchKernelGGL(kern, dim3((P.B + GS - 1) / GS), dim3(GS * kThreads), lds, stream, P, S, A);
    mfma_last_group() = GS;
    return 0;
}

// Four signals per workgroup pay off once a CU would otherwise hold more than two signals in turn (B > 2 x CUs):
// one round of four overlapping signals per CU instead of two rounds of two.  HSCMP_MFMA_QUAD=0/1 forces the choice
// (tests run both; the results are bit-identical).  Returns the signals per workgroup that `dry == false` launched.
template <typename Tile, int S4C, bool HAS_W>
static int mfma_launch_iterate_t(hipStream_t stream, const DevParams& P, const State<typename Tile::R>& S,
                                 const MfmaArgsT<typename Tile::R>& A, bool dry = false)
{
    bool quad = false;
    if constexpr (sizeof(typename Tile::R) == 4 && S4C > 0) {
        quad = P.B > 2 * mfma_device_cus();
        if (const char* e = getenv("HSCMP_MFMA_QUAD")) quad = atoi(e) != 0;
        if (quad && mfma_launch_iterate_g<Tile, S4C, HAS_W, 4>(stream, P, S, A, dry) == 0) return 0;
    }
    return mfma_launch_iterate_g<Tile, S4C, HAS_W, 1>(stream, P, S, A, dry);
}

template <typename R> inline MfmaArgsT<R> mfma_args(const DevParams& P, const State<R>& S, const R* dimg)
{
    using Tile = typename TileOf<R>::type;
    MfmaArgsT<R> A;
    A.dimg = dimg; A.G = Tile::groups(P.K); A.S4 = mfma_chunks(P.W); A.has_w = S.weights != nullptr;
    return A;
}

#define HSCMP_MFMA_DISPATCH(FN)                                                                       \
    do {                                                                                              \
        const bool hw = A.has_w != 0;                                                                 \
        switch (A.S4) {                                                                               \
        case 8: return hw ? FN<Tile, 8, true>(stream, P, S, A, dry) : FN<Tile, 8, false>(stream, P, S, A, dry);  \
        case 4: return hw ? FN<Tile, 4, true>(stream, P, S, A, dry) : FN<Tile, 4, false>(stream, P, S, A, dry);  \
        case 2: return hw ? FN<Tile, 2, true>(stream, P, S, A, dry) : FN<Tile, 2, false>(stream, P, S, A, dry);  \
        default: return hw ? FN<Tile, 0, true>(stream, P, S, A, dry) : FN<Tile, 0, false>(stream, P, S, A, dry); \
        }                                                                                             \
    } while (0)

// dry: only check that the kernel can be configured for this shape (LDS attribute), queue nothing
template <typename R> inline int mfma_launch_corr_init(hipStream_t stream, const DevParams& P, const State<R>& S, const R* dimg, bool dry = false)
{
    using Tile = typename TileOf<R>::type;
    const MfmaArgsT<R> A = mfma_args<R>(P, S, dimg);
    HSCMP_MFMA_DISPATCH(mfma_launch_corr_init_t);
}

template <typename R> inline int mfma_launch_iterate(hipStream_t stream, const DevParams& P, const State<R>& S, const R* dimg, bool dry = false)
{
    using Tile = typename TileOf<R>::type;
    const MfmaArgsT<R> A = mfma_args<R>(P, S, dimg);
    HSCMP_MFMA_DISPATCH(mfma_launch_iterate_t);
}

}  // namespace hscmp
