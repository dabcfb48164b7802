// hscmp_sparse.h -- sparsity-aware correlation for multi-feature inputs (F > 1), any dtype.
//
// Levels >= 1 of the hierarchical encoder (hsc/modeling.py:1427-1492) run the same matching
// pursuit on the previous level's coefficient matrix [T, K_prev]: F = K_prev features, of which
// only a handful per window are non-zero.  The reference correlates it densely (a [T, F*W] window
// matrix, modeling.py:181-187: 2*T*K*W*F flop).  Here the non-zeros of the residual rows that a
// block of output rows can see are gathered into LDS, sorted by (feature, row), and every output
// c[t,k] runs the pinned fma chain over the gathered entries only.  A zero sample contributes
// fma(0, d, acc) == acc to the dense chain, so skipping it is exact (up to the sign of a zero
// result); the (f outer, w inner) order of the pinned chain is the (feature, row) sort order.
// Windows that are not sparse (more than kNzMax entries) fall back to the dense chain.
//
// Used for both the initial correlation (zero padded, modeling.py:1077) -- only rows within reach
// of a non-zero input row can be non-zero -- and the local re-correlation after each atom
// (reflect padded, modeling.py:1018-1051).
#pragma once

#include "hscmp_kernels.h"

namespace hscmp {

constexpr int kNzMax = 1024;            // gathered non-zeros per row block (LDS list)

template <typename R> struct SparseArgs {
    const R* Dt;        // dictionary transposed to [W][F][K] (coalesced over atoms)
    R* scratch;         // [B][(2W-1)*K] per-workgroup table of the rows being recomputed
};

template <typename R> struct SparseLds {
    R val[kNzMax];
    int key[kNzMax];    // (f << 16) | local row j
    R sval[kNzMax];     // sorted copies
    int skey[kNzMax];
    int count;
    int wtot[kWaves];
};

// Dense chain for output rows [row0, row0+nrows) (global positions), all atoms: the generic
// fallback.  reflect: np.pad 'reflect' w.r.t. the slice [sidx, sidx+nslice) (modeling.py:1046);
// otherwise zero padding (modeling.py:159-164).  Results go to the per-workgroup table tab[row][k].
template <typename R>
__device__ __forceinline__ void dense_rows_to_table(const DevParams& P, const R* __restrict__ r, const R* __restrict__ D,
                                                    int row0, int nrows, bool reflect, int sidx, int nslice, R* tab)
{
    const int T = P.T, K = P.K, W = P.W, F = P.F;
    for (int o = threadIdx.x; o < nrows * K; o += kThreads) {
        const int row = o / K, k = o - row * K;
        const int t = row0 + row;
        R acc = (R)0;
        if (t >= 0 && t < T) {
            const R* dk = D + (int64_t)k * W * F;
            for (int f = 0; f < F; ++f)
                for (int w = 0; w < W; ++w) {
                    int g = t - P.off + w;
                    R xv;
                    if (reflect) { g = reflect_index(g, sidx, nslice); xv = r[(int64_t)g * F + f]; }
                    else xv = (g >= 0 && g < T) ? r[(int64_t)g * F + f] : (R)0;
                    acc = rfma(xv, dk[w * F + f], acc);
                }
        }
        tab[o] = acc;
    }
}

// Rows [row0, row0+nrows) x all atoms -> per-position best (best_c, best_k), sparsity aware.
//   nrows <= 2W-1 (table capacity).  All threads of the workgroup call it (contains barriers).
template <typename R>
__device__ __forceinline__ void sparse_rows(const DevParams& P, const State<R>& S, const Sig<R>& G, const SparseArgs<R>& A,
                                            SparseLds<R>& L, int row0, int nrows, bool reflect, int sidx, int nslice)
{
    const int T = P.T, K = P.K, W = P.W, F = P.F, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    R* tab = A.scratch + (int64_t)blockIdx.x * (2 * W - 1) * K;
    const int nwin = nrows + W - 1;                 // residual rows the block can see
    const int g0 = row0 - P.off;                    // global row of local row 0

    // ---- 1. gather the non-zeros of the window, row-major (coalesced), order-preserving
    if (tid == 0) L.count = 0;
    __syncthreads();
    const int total = nwin * F;
    int running = 0;
    bool overflow = false;
    for (int base = 0; base < total; base += kThreads) {
        const int e = base + tid;
        R v = (R)0;
        int j = 0, f = 0;
        if (e < total) {
            j = e / F; f = e - j * F;
            int g = g0 + j;
            if (reflect) { g = reflect_index(g, sidx, nslice); v = G.r[(int64_t)g * F + f]; }
            else if (g >= 0 && g < T) v = G.r[(int64_t)g * F + f];
        }
        const int flag = (v != (R)0) ? 1 : 0;
        const unsigned long long mask = __ballot(flag);
        const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) L.wtot[wv] = __popcll(mask);
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < kWaves; ++q) { const int c = L.wtot[q]; if (q < wv) before += c; tot += c; }
        if (flag) {
            const int o = running + before + prefix;
            if (o < kNzMax) { L.val[o] = v; L.key[o] = (f << 16) | j; }
        }
        running += tot;
        if (running > kNzMax) { overflow = true; break; }       // uniform
    }
    __syncthreads();

    if (overflow) {
        dense_rows_to_table(P, G.r, S.D, row0, nrows, reflect, sidx, nslice, tab);
    } else {
        // ---- 2. sort by (feature, row): the pinned chain order f outer / w inner (rank sort)
        const int n = running;
        for (int i = tid; i < n; i += kThreads) {
            const int key = L.key[i];
            int rank = 0;
            for (int q = 0; q < n; ++q) rank += (L.key[q] < key) ? 1 : 0;      // keys are distinct
            L.skey[rank] = key; L.sval[rank] = L.val[i];
        }
        __syncthreads();
        // ---- 3. every (row, atom): chain over the entries inside the row's window
        for (int o = tid; o < nrows * K; o += kThreads) {
            const int row = o / K, k = o - row * K;
            R acc = (R)0;
            for (int i = 0; i < n; ++i) {
                const int key = L.skey[i];
                const int w = (key & 0xffff) - row;
                if (w >= 0 && w < W) acc = rfma(L.sval[i], A.Dt[((int64_t)w * F + (key >> 16)) * K + k], acc);
            }
            tab[o] = acc;
        }
    }
    __syncthreads();
    // ---- 4. per-row best over atoms (first k wins ties), one wave per row
    for (int row = wv; row < nrows; row += kWaves) {
        const int t = row0 + row;
        if (t < 0 || t >= T) continue;                           // overlapReplace clipping (utils.py:133-161)
        Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
        for (int k = lane; k < K; k += 64) {
            const R sc = score_of(tab[row * K + k], k, S.weights);
            if (sc > best.s) { best.s = sc; best.i = k; }
        }
        best = wave_argmax(best);
        if (lane == 0) { G.bc[t] = tab[row * K + best.i]; G.bk[t] = best.i; }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// policy of iterate_kernel: local re-correlation of the 2W-1 touched rows, reflect padded
// ------------------------------------------------------------------------------------------------
template <typename R> struct SparseRecorr {
    static constexpr int kMaxSegments = kMaxSeg;
    static constexpr bool kFused = false;
    static constexpr bool kScoreOnly = false;
    using Shared = IterSharedT<R, kMaxSeg>;
    using Args = SparseArgs<R>;
    static size_t extra_lds_bytes(const DevParams&) { return sizeof(SparseLds<R>); }
    static __device__ __forceinline__ void prologue(const DevParams&, const State<R>&, const Args&, char*) {}
    static __device__ __forceinline__ void epilogue(const DevParams&, const State<R>&, const Args&, char*) {}
    static __device__ __forceinline__ void resolve_wave(const DevParams&, const State<R>&, const Sig<R>&, const Args&, char*,
                                                        int, int, int&, R&) {}
    template <typename SH>
    static __device__ __forceinline__ void run(const DevParams& P, const State<R>& S, const Sig<R>& G, SH&, const Args& A,
                                               char* lds, int p)
    {
        SparseLds<R>& L = *reinterpret_cast<SparseLds<R>*>(lds);
        const int T = P.T, W = P.W;
        const int tstart = p - P.off - (W - 1);            // :1028-1033
        const int tend = p + W / 2 + (W - 1);              // :1038
        const int sidx = tstart < 0 ? 0 : tstart;          // :1034
        const int eidx = tend > T - 1 ? T - 1 : tend;      // :1039
        sparse_rows(P, S, G, A, L, p - (W - 1), 2 * W - 1, true, sidx, eidx - sidx + 1);
    }
};

// ------------------------------------------------------------------------------------------------
// initial correlation of a sparse multi-feature input (modeling.py:1077): only rows whose window
// contains a non-zero input row can be non-zero; everything else is (c = 0, k = 0).
//   grid = B, block = kThreads;  dynamic LDS = SparseLds<R> + T bits of row flags
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(kThreads) void corr_init_sparse_kernel(DevParams P, State<R> S, SparseArgs<R> A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SparseLds<R>& L = *reinterpret_cast<SparseLds<R>*>(smem);
    unsigned* rowbits = reinterpret_cast<unsigned*>(smem + ((sizeof(SparseLds<R>) + 15) / 16) * 16);   // [ceil(T/32)]
    const int b = blockIdx.x, tid = threadIdx.x, T = P.T, W = P.W, F = P.F;
    Sig<R> G{};
    G.r = S.residual + (int64_t)b * T * F;
    G.bc = S.best_c + (int64_t)b * T;
    G.bk = S.best_k + (int64_t)b * T;
    const int nwords = (T + 31) / 32;
    for (int i = tid; i < nwords; i += kThreads) rowbits[i] = 0u;
    for (int t = tid; t < T; t += kThreads) { G.bc[t] = (R)0; G.bk[t] = 0; }
    __syncthreads();
    // input rows with a non-zero sample
    for (int64_t e = tid; e < (int64_t)T * F; e += kThreads)
        if (G.r[e] != (R)0) atomicOr(&rowbits[(int)(e / F) >> 5], 1u << ((int)(e / F) & 31));
    __syncthreads();
    // walk the output rows in blocks of up to 2W-1; a block is computed iff some input row in its reach is set
    const int blk = 2 * W - 1;
    for (int row0 = 0; row0 < T; row0 += blk) {
        const int nrows = min(blk, T - row0);
        const int lo = max(0, row0 - P.off), hi = min(T - 1, row0 + nrows - 1 - P.off + W - 1);
        int any = 0;
        for (int t = lo + tid; t <= hi; t += kThreads) any |= (rowbits[t >> 5] >> (t & 31)) & 1u;
        if (__syncthreads_or(any)) sparse_rows(P, S, G, A, L, row0, nrows, false, 0, 0);
    }
}

}  // namespace hscmp
