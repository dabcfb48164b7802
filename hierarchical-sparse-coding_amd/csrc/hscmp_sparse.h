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
// Windows that are not sparse (more entries than the LDS list holds) fall back to the dense chain.
//
// Level dictionaries built from decompositions (hsc/dataset.py:137-194, a few lower-level events per
// atom plus the unit "singleton" atoms of :826-860) are themselves almost all zero.  For those only
// the non-zero products are formed: every gathered input non-zero (f, j) is paired with the
// dictionary non-zeros (k, w) of feature f, the pairs are sorted by (output, chain order) and one
// thread per output runs its chain.  All other outputs are +0.
//
// Finding the window's non-zeros is the expensive part (a window is (3W-2) x F values, 400 KB at
// BASELINE config 5 level 2).  With a sparse dictionary the set of (row, feature) cells that can be
// non-zero is tracked explicitly: per-row feature lists in global memory, filled from the input and
// extended by the few cells every subtracted atom touches; the gather reads the lists and then exactly
// those cells (two dependent round trips).  Rows whose list overflows are read densely.  Without
// dictionary lists a per-signal row-occupancy bitmap at least skips the all-zero rows.
//
// Used for both the initial correlation (zero padded, modeling.py:1077) -- only rows within reach
// of a non-zero input row can be non-zero -- and the local re-correlation after each atom
// (reflect padded, modeling.py:1018-1051).
#pragma once

#include "hscmp_kernels.h"

namespace hscmp {

// Capacities of the LDS lists of one row block, chosen per dictionary shape on the host (sparse_caps):
struct SparseCaps {
    int nz;     // gathered non-zeros of the window
    int rec;    // (input non-zero, dictionary non-zero) pairs; nz <= rec <= 1024 (10-bit index in the row arg-max)
    int rows;   // occupied rows of the window (also the largest row count the bucketed pair sort handles)
    int ewaves; // waves that take weak-atom-filter energies at the same time (SparseRecorr::wave_window_listed's table)
};
inline SparseCaps sparse_caps(int W, bool packed = false)
{
    // a window spans 3W-2 rows; room for about 2 input non-zeros and 8 pairs per row, within 128..512 / 256..1024
    // (packed: the four-workgroups-per-CU form of the loop, half the pair capacity -- overflows take the per-atom chains)
    auto pow2 = [](int v) { int p = 1; while (p < v) p <<= 1; return p; };
    SparseCaps c;
    c.nz = std::min(512, std::max(128, pow2(2 * (3 * W - 2))));
    c.rec = std::min(1024, std::max(256, pow2((packed ? 4 : 8) * (3 * W - 2))));
    c.rows = 256;
    c.ewaves = packed ? kWaves / 2 : kWaves;
    return c;
}
constexpr int kDictListMaxPerAtom = 32; // a dictionary with at most this many non-zeros per atom (average) gets per-atom lists
constexpr int kRowBitsMaxT = 262144;    // longest signal with a row-occupancy bitmap in LDS (32 KB)

template <typename R> struct SparseArgs {
    const R* Dt;        // dictionary transposed to [W][F][K] (coalesced over atoms)
    R* scratch;         // [B][(2W-1)*K] per-workgroup table of the rows being recomputed
    unsigned char* rowflag;   // [B][T] 1 = the residual row may hold a non-zero (nullptr: no bitmap, T too long)
    int rowflag_filled; // the flags of the input were written by the caller (level chaining): no scan of the input
    // Non-zeros of a sparse dictionary, per atom in chain order (f outer, w inner); nullptr when the
    // dictionary is not sparse enough (then the window's non-zeros meet the dense dictionary).
    const int* nzptr;   // [K+1]
    const int* nzwf;    // [nnz] (w << 16) | f
    const R* nzval;     // [nnz]
    // the same non-zeros grouped by feature (for the sparse-input x sparse-dictionary pairing)
    const int* fptr;    // [F+1]
    const int* fkw;     // [nnz] (k << 16) | w
    const R* fval;      // [nnz]
    // Per-row lists of the features that may be non-zero in the residual (sparse dictionaries only): the
    // window gather reads them instead of scanning rows of F values.  nullptr: scan (row bitmap).
    int* rl_cnt;        // [B][T] members of row t (> rl_cap: list overflowed, the row is read densely)
    int* rl_f;          // [B][T][rl_cap] feature indices, -1 = empty
    int rl_cap;         // power of two
    int rl_filled;      // the caller (level chaining) has already listed the input's non-zeros
    SparseCaps caps;    // LDS list capacities
    int nnz;            // non-zeros of the dictionary (length of the lists)
    const R* wts;       // atom weights (State::weights, or their LDS copy)
};

constexpr int kDictLdsBytes = 12288;    // the by-feature lists are copied to LDS when they fit this ...
constexpr int kWeightLdsBytes = 8192;   // ... and so are the atom weights

template <typename R> __host__ __device__ inline size_t staged_weight_bytes(const DevParams& P, const SparseArgs<R>& A)
{
    const size_t b = (size_t)P.K * sizeof(R);
    return (A.wts && b <= (size_t)kWeightLdsBytes) ? (b + 15) / 16 * 16 : 0;
}
template <typename R> __host__ __device__ inline size_t staged_list_bytes(const DevParams& P, const SparseArgs<R>& A)
{
    if (!A.fptr) return 0;
    const size_t b = (((size_t)(P.F + 1) + (size_t)A.nnz) * sizeof(int) + 7) / 8 * 8 + (size_t)A.nnz * sizeof(R);
    return b <= (size_t)kDictLdsBytes ? (b + 15) / 16 * 16 : 0;
}
// (the per-atom list offsets: the residual update starts from them, K+1 ints)
template <typename R> __host__ __device__ inline size_t staged_ptr_bytes(const DevParams& P, const SparseArgs<R>& A)
{
    const size_t b = ((size_t)P.K + 1) * sizeof(int);
    return (A.nzptr && b <= (size_t)kWeightLdsBytes) ? (b + 15) / 16 * 16 : 0;
}
// LDS bytes of the staged weights + by-feature lists + per-atom offsets (what does not fit stays in global memory)
template <typename R> __host__ __device__ inline size_t staged_dict_bytes(const DevParams& P, const SparseArgs<R>& A)
{
    return staged_weight_bytes(P, A) + staged_list_bytes(P, A) + staged_ptr_bytes(P, A);
}

// Arguments whose lists / weights point at their LDS copies at `base` (unchanged for what is not staged).
template <typename R>
__device__ __forceinline__ SparseArgs<R> dict_view(const DevParams& P, const SparseArgs<R>& A, char* base)
{
    SparseArgs<R> B = A;
    const size_t wb = staged_weight_bytes(P, A);
    if (wb) B.wts = reinterpret_cast<const R*>(base);
    if (staged_list_bytes(P, A)) {
        int* fptr = reinterpret_cast<int*>(base + wb);
        B.fptr = fptr; B.fkw = fptr + (P.F + 1);
        B.fval = reinterpret_cast<const R*>(base + wb + (((size_t)(P.F + 1) + (size_t)A.nnz) * sizeof(int) + 7) / 8 * 8);
    }
    if (staged_ptr_bytes(P, A)) B.nzptr = reinterpret_cast<const int*>(base + wb + staged_list_bytes(P, A));
    return B;
}

// Copy weights / lists to LDS at `base` (all threads; the caller synchronises).
template <typename R>
__device__ __forceinline__ void stage_dict(const DevParams& P, const SparseArgs<R>& A, char* base)
{
    const SparseArgs<R> B = dict_view(P, A, base);
    if (staged_weight_bytes(P, A)) for (int i = threadIdx.x; i < P.K; i += kThreads) const_cast<R*>(B.wts)[i] = A.wts[i];
    if (staged_list_bytes(P, A)) {
        for (int i = threadIdx.x; i <= P.F; i += kThreads) const_cast<int*>(B.fptr)[i] = A.fptr[i];
        for (int i = threadIdx.x; i < A.nnz; i += kThreads) { const_cast<int*>(B.fkw)[i] = A.fkw[i]; const_cast<R*>(B.fval)[i] = A.fval[i]; }
    }
    if (staged_ptr_bytes(P, A)) for (int i = threadIdx.x; i <= P.K; i += kThreads) const_cast<int*>(B.nzptr)[i] = A.nzptr[i];
}

// LDS lists of one row block: a view into the policy's dynamic LDS (sizes = SparseCaps).
template <typename R> struct SparseLds {
    int* ctl;                   // [0] gathered non-zeros, [1] occupied window rows, [2] pairs
    R* val; int* key;           // [nz] gathered non-zeros: value, (f << 16) | local row j
    int* rowj; int* rowg;       // [rows+1] occupied rows of the window: local row, global (reflected) row;
                                //          reused by the pair sort as per-row counts / start offsets
    // pairing of a sparse window with a sparse dictionary: one record per non-zero product
    unsigned long long* rkey;   // [rec] row << 48 | k << 32 | f << 16 | w : output first, then chain order
    R* rx; R* rd;               // [rec] the two factors
    R* out;                     // [rec] chain result, at the chain's first sorted position
    int* perm;                  // [rec] sorted position -> record
    unsigned* okey;             // [rec] per sorted position: row << 16 | k if it starts an output's chain, else ~0
    // gathered window x dense dictionary (never used together with the pairing): sorted copies
    R* sval; int* skey;         // [nz], aliases of rx / perm
    SparseCaps caps;
};

__host__ __device__ inline size_t sparse_lds_bytes_of(const SparseCaps& c, size_t es)
{
    size_t b = 16 + (size_t)c.nz * (es + 4) + (size_t)(c.rows + 1) * 8 + 8 + (size_t)c.rec * (8 + 3 * es + 8);
    const size_t wave_energy = (size_t)c.ewaves * kThreads * (2 * es + 4);       // SparseRecorr::wave_window_listed borrows the region
    if (b < wave_energy) b = wave_energy;
    return (b + 15) / 16 * 16;
}
template <typename R> __host__ __device__ inline size_t sparse_lds_bytes(const SparseCaps& c) { return sparse_lds_bytes_of(c, sizeof(R)); }

template <typename R> __device__ __forceinline__ SparseLds<R> sparse_lds_view(char* base, const SparseCaps& c)
{
    SparseLds<R> L;
    L.caps = c;
    L.ctl = reinterpret_cast<int*>(base);
    char* p = base + 16;
    L.rkey = reinterpret_cast<unsigned long long*>(p); p += (size_t)c.rec * 8;
    L.rx = reinterpret_cast<R*>(p); p += (size_t)c.rec * sizeof(R);
    L.rd = reinterpret_cast<R*>(p); p += (size_t)c.rec * sizeof(R);
    L.out = reinterpret_cast<R*>(p); p += (size_t)c.rec * sizeof(R);
    L.val = reinterpret_cast<R*>(p); p += (size_t)c.nz * sizeof(R);
    p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 7) & ~(uintptr_t)7);
    L.perm = reinterpret_cast<int*>(p); p += (size_t)c.rec * 4;
    L.okey = reinterpret_cast<unsigned*>(p); p += (size_t)c.rec * 4;
    L.key = reinterpret_cast<int*>(p); p += (size_t)c.nz * 4;
    L.rowj = reinterpret_cast<int*>(p); p += (size_t)(c.rows + 1) * 4;
    L.rowg = reinterpret_cast<int*>(p);
    L.sval = L.rx; L.skey = L.perm;
    return L;
}

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

// The list counters are advanced with atomics (performed in L2): read them past the vector L1 as well.
__device__ __forceinline__ int list_count(const int* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Non-zeros of the residual rows [g0, g0+nwin) (reflected / zero padded) -> L.val / L.key, in any
// order.  With per-row feature lists (A.rl_cnt) only the listed cells are read; otherwise rows are scanned,
// and with rowbits (LDS, may be nullptr) only those whose bit is set.  Returns the number
// of non-zeros found (> caps.nz: the list overflowed and is unusable).  Contains barriers.
template <typename R, bool LAUNDER = false>
__device__ __forceinline__ int gather_window(const DevParams& P, const Sig<R>& G, const SparseArgs<R>& A, const SparseLds<R>& L,
                                             const unsigned* rowbits, int g0, int nwin, bool reflect, int sidx, int nslice)
{
    const int T = P.T, F = P.F, tid = laundered_tid<LAUNDER>(), lane = tid & 63, wv = tid >> 6;
    const int nzcap = L.caps.nz, rowcap = L.caps.rows;
    if (tid == 0) { L.ctl[0] = 0; L.ctl[1] = 0; L.ctl[2] = 0; L.ctl[3] = 0; }
    __syncthreads();
    if (A.rl_cnt) {
        // listed cells: items = (window row, list slot); first the feature indices, then the values
        const int C = A.rl_cap, shift = __ffs(C) - 1;
        const int* cnt = A.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A.rl_f + (int64_t)blockIdx.x * T * C;
        constexpr int kV = 4;
        const int items = C == 8 ? 0 : nwin << shift;
        if (C == 8) {
            // one thread per window row: its count and its list in one round trip, its cells in a second
            for (int j = tid; j < nwin; j += kThreads) {
                int gg = g0 + j;
                bool ok = true;
                if (reflect) gg = reflect_index(gg, sidx, nslice);
                else ok = gg >= 0 && gg < T;
                if (!ok) continue;
                const int n = list_count(cnt + gg);
                const int4* row = reinterpret_cast<const int4*>(lf + (int64_t)gg * 8);
                const int4 a = row[0], b = row[1];
                if (n > C) {                              // overflowed list: the row goes to the dense scan below
                    const int o = atomicAdd(&L.ctl[1], 1);
                    if (o < rowcap) { L.rowj[o] = j; L.rowg[o] = gg; }
                    continue;
                }
                if (n <= 0) continue;
                const int fs[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                R vs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vs[u] = fs[u] >= 0 ? G.r[(int64_t)gg * F + fs[u]] : (R)0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (vs[u] != (R)0) {
                        const int o = atomicAdd(&L.ctl[0], 1);
                        if (o < nzcap) { L.val[o] = vs[u]; L.key[o] = (fs[u] << 16) | j; }
                    }
            }
        }
        for (int it0 = tid; it0 < items; it0 += kThreads * kV) {
            int g[kV], f[kV], j[kV], n[kV];
#pragma unroll
            for (int u = 0; u < kV; ++u) {
                const int it = it0 + u * kThreads;
                f[u] = -1; n[u] = 0; g[u] = 0; j[u] = it >> shift;
                if (it < items) {
                    int gg = g0 + j[u];
                    bool ok = true;
                    if (reflect) gg = reflect_index(gg, sidx, nslice);
                    else ok = gg >= 0 && gg < T;
                    if (ok) { g[u] = gg; n[u] = list_count(cnt + gg); f[u] = lf[((int64_t)gg << shift) + (it & (C - 1))]; }
                }
            }
            R v[kV];
#pragma unroll
            for (int u = 0; u < kV; ++u) {
                v[u] = (R)0;
                if (n[u] > C) {                       // overflowed list: the row goes to the dense scan below (once)
                    if (((it0 + u * kThreads) & (C - 1)) == 0) {
                        const int o = atomicAdd(&L.ctl[1], 1);
                        if (o < rowcap) { L.rowj[o] = j[u]; L.rowg[o] = g[u]; }
                    }
                } else if (f[u] >= 0) v[u] = G.r[(int64_t)g[u] * F + f[u]];
            }
#pragma unroll
            for (int u = 0; u < kV; ++u)
                if (v[u] != (R)0) {
                    const int o = atomicAdd(&L.ctl[0], 1);
                    if (o < nzcap) { L.val[o] = v[u]; L.key[o] = (f[u] << 16) | j[u]; }
                }
        }
    } else {
        // rows that exist (zero padding) and may hold a non-zero
        for (int j = tid; j < nwin; j += kThreads) {
            int g = g0 + j;
            bool ok = true;
            if (reflect) g = reflect_index(g, sidx, nslice);
            else ok = g >= 0 && g < T;
            if (ok && rowbits) ok = ((rowbits[g >> 5] >> (g & 31)) & 1u) != 0;
            if (ok) {
                const int o = atomicAdd(&L.ctl[1], 1);
                if (o < rowcap) { L.rowj[o] = j; L.rowg[o] = g; }
            }
        }
    }
    __syncthreads();
    const int nr = L.ctl[1];
    if (nr > rowcap) return nzcap + 1;
    // (row, 64-feature chunk) items round-robin over the waves, eight loads in flight per lane
    constexpr int kU = 8;
    const int nchunks = (F + 63) >> 6;
    const int items = nr * nchunks;
    for (int it0 = wv; it0 < items; it0 += kWaves * kU) {
        R v[kU];
        int key[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int it = it0 + u * kWaves;
            v[u] = (R)0; key[u] = 0;
            if (it < items) {
                const int q = it / nchunks, f = ((it - q * nchunks) << 6) + lane;
                if (f < F) { v[u] = G.r[(int64_t)L.rowg[q] * F + f]; key[u] = (f << 16) | L.rowj[q]; }
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (v[u] != (R)0) {
                const int o = atomicAdd(&L.ctl[0], 1);
                if (o < nzcap) { L.val[o] = v[u]; L.key[o] = key[u]; }
            }
    }
    __syncthreads();
    return L.ctl[0];
}

__device__ __forceinline__ unsigned long long score_bits(double s) { return (unsigned long long)__double_as_longlong(s); }
__device__ __forceinline__ unsigned long long score_bits(float s) { return (unsigned long long)__float_as_uint(s); }

// Rows [row0, row0+nrows) x all atoms -> per-position best (best_c, best_k), sparsity aware.
//   nrows <= 2W-1 (table capacity).  All threads of the workgroup call it (contains barriers).
template <typename R, bool LAUNDER = false>
__device__ __forceinline__ void sparse_rows(const DevParams& P, const State<R>& S, const Sig<R>& G, const SparseArgs<R>& A,
                                            const SparseLds<R>& L, const unsigned* rowbits, int row0, int nrows, bool reflect,
                                            int sidx, int nslice, bool gathered = false)
{
    const int T = P.T, K = P.K, W = P.W, F = P.F, tid = laundered_tid<LAUNDER>(), lane = tid & 63, wv = tid >> 6;
    const int nzcap = L.caps.nz, reccap = L.caps.rec;
    R* tab = A.scratch + (int64_t)blockIdx.x * (2 * W - 1) * K;
    const int nwin = nrows + W - 1;                 // residual rows the block can see
    const int g0 = row0 - P.off;                    // global row of local row 0

    // ---- 1. the non-zeros of the window (any order)
    HSCMP_STAMP_BEGIN();
    int n;
    if (gathered) {
        // (SparseRecorr::merged_update left the window's cells in L.key / L.val; a cell the atom cancelled holds 0)
        n = L.ctl[0];
        lds_barrier();
        if (tid == 0) { L.ctl[1] = 0; L.ctl[2] = 0; L.ctl[3] = 0; }
        lds_barrier();
    } else {
        n = gather_window<R, LAUNDER>(P, G, A, L, rowbits, g0, nwin, reflect, sidx, nslice);
    }
    HSCMP_STAMP(40);
    HSCMP_TALLY(0, 1); HSCMP_TALLY(1, n); HSCMP_TALLY(2, n > nzcap); HSCMP_TALLY(5, L.ctl[1]);

    if (A.fptr && n <= nzcap) {
        // ---- sparse window x sparse dictionary: only the non-zero products are formed.
        //  (every barrier of this branch orders LDS lists only -- lds_barrier: the residual / row-list stores of the
        //   update that went before keep draining underneath)
        //  a. pair every input non-zero (f, j) with the dictionary non-zeros (k, w) of feature f: output row j - w
        //     (all (input non-zero, list entry) combinations side by side: items = non-zeros x the longest list among them)
        int* lbeg = L.perm; int* llen = reinterpret_cast<int*>(L.okey);        // [n] (both are free until the sort)
        for (int i = tid; i < n; i += kThreads) {
            const int f = L.key[i] >> 16;
            const int b = A.fptr[f], len = (L.val[i] != (R)0) ? A.fptr[f + 1] - b : 0;     // (a zero factor adds nothing)
            lbeg[i] = b; llen[i] = len;
            atomicMax(&L.ctl[3], len);
        }
        lds_barrier();
        const int longest = L.ctl[3];
        for (int it = tid; it < n * longest; it += kThreads) {
            const int i = it / longest, sl = it - i * longest;
            if (sl >= llen[i]) continue;
            const int e = lbeg[i] + sl;
            const int key = L.key[i], f = key >> 16, j = key & 0xffff;
            const int kw = A.fkw[e], w = kw & 0xffff, row = j - w;
            const R d = A.fval[e];
            if (row < 0 || row >= nrows) continue;
            const int t = row0 + row;
            if (t < 0 || t >= T) continue;
            const int o = atomicAdd(&L.ctl[2], 1);
            if (o < reccap) {
                L.rkey[o] = ((unsigned long long)row << 48) | ((unsigned long long)((unsigned)kw >> 16) << 32) |
                            ((unsigned long long)f << 16) | (unsigned)w;
                L.rx[o] = L.val[i]; L.rd[o] = d;
            }
        }
        lds_barrier();
        const int m = L.ctl[2];
        HSCMP_STAMP(41);
        HSCMP_TALLY(3, m); HSCMP_TALLY(4, m > reccap);
        if (m <= reccap) {
            //  b. sort by (output, chain order); the keys are distinct.  Few output rows: bucket by row first
            //     (counts -> offsets -> members), then rank inside the bucket; otherwise one rank sort over all.
            const bool bucketed = nrows <= L.caps.rows && m > 64;
            int* cnt = L.rowj;                       // [nrows]   (the window-row lists are dead by now)
            int* start = L.rowg;                     // [nrows+1]
            if (bucketed) {
                for (int r = tid; r < nrows; r += kThreads) cnt[r] = 0;
                lds_barrier();
                for (int i = tid; i < m; i += kThreads) atomicAdd(&cnt[(int)(L.rkey[i] >> 48)], 1);
                lds_barrier();
                if (wv == 0) {
                    // exclusive prefix sum over the rows by one wave: each lane owns a run of rows
                    const int per = (nrows + 63) >> 6;
                    int local = 0;
                    for (int q = 0; q < per; ++q) { const int r = lane * per + q; if (r < nrows) local += cnt[r]; }
                    int incl = local;
                    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                    int run = incl - local;
                    for (int q = 0; q < per; ++q) { const int r = lane * per + q; if (r < nrows) { start[r] = run; run += cnt[r]; } }
                    if (lane == 63) start[nrows] = incl;
                }
                lds_barrier();
                for (int r = tid; r < nrows; r += kThreads) cnt[r] = 0;          // cursors
                lds_barrier();
                unsigned* member = L.okey;           // bucket members (record indices); okey proper is written in step c
                for (int i = tid; i < m; i += kThreads) {
                    const int r = (int)(L.rkey[i] >> 48);
                    member[start[r] + atomicAdd(&cnt[r], 1)] = (unsigned)i;
                }
                lds_barrier();
                for (int i = tid; i < m; i += kThreads) {
                    const unsigned long long key = L.rkey[i];
                    const int r = (int)(key >> 48), b0 = start[r], b1 = start[r + 1];
                    int rank = 0;
                    for (int q = b0; q < b1; ++q) rank += (L.rkey[member[q]] < key) ? 1 : 0;
                    L.perm[b0 + rank] = i;
                }
            } else {
                for (int i = tid; i < m; i += kThreads) {
                    const unsigned long long key = L.rkey[i];
                    int rank = 0;
                    int q = 0;
                    for (; q + 8 <= m; q += 8) {                         // eight keys per LDS round trip
                        unsigned long long kq[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) kq[u] = L.rkey[q + u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) rank += (kq[u] < key) ? 1 : 0;
                    }
                    for (; q < m; ++q) rank += (L.rkey[q] < key) ? 1 : 0;
                    L.perm[rank] = i;
                }
            }
            lds_barrier();
            HSCMP_STAMP(42);
            //  c. one chain per output, run by the thread of its first record (f outer, w inner from +0)
            for (int sp = tid; sp < m; sp += kThreads) {
                const unsigned ok = (unsigned)(L.rkey[L.perm[sp]] >> 32);
                unsigned mark = ~0u;
                if (sp == 0 || (unsigned)(L.rkey[L.perm[sp - 1]] >> 32) != ok) {
                    R acc = (R)0;
                    for (int q = sp; q < m; ++q) {
                        const int rec = L.perm[q];
                        if ((unsigned)(L.rkey[rec] >> 32) != ok) break;
                        acc = rfma(L.rx[rec], L.rd[rec], acc);
                    }
                    L.out[sp] = acc;
                    mark = ok;
                }
                L.okey[sp] = mark;
            }
            lds_barrier();
            HSCMP_STAMP(43);
            //  d. per-row best over atoms: the listed outputs against the zeros of all the others -- a zero score never
            //     beats k = 0, the first of the ties; among equal scores the lowest atom wins (np.argmax).
            if (nrows <= reccap) {
                // all listed outputs side by side: row maximum of the score bits (scores are >= 0, so their bit
                // patterns order like the values), then the lowest atom that reaches it, then its coefficient.
                // The per-row cells take the place of the factor / key arrays, which are dead after step c.
                unsigned long long* rmax = L.rkey;
                int* rk = L.perm;
                R* c0 = L.rx; R* rc = L.rd;
                for (int row = tid; row < nrows; row += kThreads) { rmax[row] = 0ull; rk[row] = INT_MAX; c0[row] = (R)0; }
                lds_barrier();
                for (int sp = tid; sp < m; sp += kThreads) {
                    const unsigned ok = L.okey[sp];
                    if (ok == ~0u) continue;
                    const int row = (int)(ok >> 16), kk = (int)(ok & 0xffffu);
                    const R o = L.out[sp];
                    if (kk == 0) c0[row] = o;                            // the value of the default winner
                    const R sc = score_of(o, kk, A.wts);
                    if (sc > (R)0) atomicMax(&rmax[row], score_bits(sc));
                }
                lds_barrier();
                for (int sp = tid; sp < m; sp += kThreads) {
                    const unsigned ok = L.okey[sp];
                    if (ok == ~0u) continue;
                    const int row = (int)(ok >> 16), kk = (int)(ok & 0xffffu);
                    const R sc = score_of(L.out[sp], kk, A.wts);
                    if (sc > (R)0 && score_bits(sc) == rmax[row]) atomicMin(&rk[row], kk);
                }
                lds_barrier();
                for (int sp = tid; sp < m; sp += kThreads) {
                    const unsigned ok = L.okey[sp];
                    if (ok == ~0u) continue;
                    const int row = (int)(ok >> 16), kk = (int)(ok & 0xffffu);
                    const R o = L.out[sp];
                    const R sc = score_of(o, kk, A.wts);
                    if (sc > (R)0 && score_bits(sc) == rmax[row] && kk == rk[row]) rc[row] = o;
                }
                lds_barrier();
                for (int row = tid; row < nrows; row += kThreads) {
                    const int t = row0 + row;
                    if (t < 0 || t >= T) continue;                       // overlapReplace clipping (utils.py:133-161)
                    const int k = rk[row];
                    if (k == INT_MAX) { G.bc[t] = c0[row]; G.bk[t] = 0; }
                    else { G.bc[t] = rc[row]; G.bk[t] = k; }
                }
                if (tid == 0) L.ctl[3] = -3;                             // rk / rc / c0 hold the rows' results (SparseRecorr::row_results)
            } else {
                // (more rows than list entries: one thread per row walks its bucket, ascending k inside a row)
                for (int row = tid; row < nrows; row += kThreads) {
                    const int t = row0 + row;
                    if (t < 0 || t >= T) continue;
                    const int b0 = bucketed ? start[row] : 0, b1 = bucketed ? start[row + 1] : m;
                    R bs = (R)0, c = (R)0;
                    int k = 0;
                    for (int sp = b0; sp < b1; ++sp) {
                        const unsigned ok = L.okey[sp];
                        if (ok == ~0u || (int)(ok >> 16) != row) continue;
                        const int kk = (int)(ok & 0xffffu);
                        const R o = L.out[sp];
                        if (kk == 0) c = o;
                        const R sc = score_of(o, kk, A.wts);
                        if (sc > bs) { bs = sc; k = kk; c = o; }
                    }
                    G.bc[t] = c; G.bk[t] = k;
                }
            }
            HSCMP_STAMP(51);
            lds_barrier();       // (LDS lists: free for the next use.  The rows' global stores drain at the caller's next full barrier)
            HSCMP_STAMP(44);
            return;
        }
    }

    if (A.nzptr) {
        // sparse dictionary, window or pair list too long: each output walks its atom's non-zeros (already in chain order)
        for (int o = tid; o < nrows * K; o += kThreads) {
            const int row = o / K, k = o - row * K;
            const int t = row0 + row;
            R acc = (R)0;
            if (t >= 0 && t < T) {
                const int e1 = A.nzptr[k + 1];
                for (int e = A.nzptr[k]; e < e1; ++e) {
                    const int wf = A.nzwf[e];
                    int g = g0 + row + (wf >> 16);
                    R xv;
                    if (reflect) { g = reflect_index(g, sidx, nslice); xv = G.r[(int64_t)g * F + (wf & 0xffff)]; }
                    else xv = (g >= 0 && g < T) ? G.r[(int64_t)g * F + (wf & 0xffff)] : (R)0;
                    acc = rfma(xv, A.nzval[e], acc);
                }
            }
            tab[o] = acc;
        }
    } else if (n > nzcap) {
        dense_rows_to_table(P, G.r, S.D, row0, nrows, reflect, sidx, nslice, tab);
    } else {
        // ---- 2. sort by (feature, row): the pinned chain order f outer / w inner (rank sort)
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
            const R sc = score_of(tab[row * K + k], k, A.wts);
            if (sc > best.s) { best.s = sc; best.i = k; }
        }
        best = wave_argmax(best);
        if (best.i == INT_MAX) best.i = 0;                       // every score of the row is NaN (a diverged pursuit): np.argmax gives 0
        if (lane == 0) { G.bc[t] = tab[row * K + best.i]; G.bk[t] = best.i; }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// policy of iterate_kernel: local re-correlation of the 2W-1 touched rows, reflect padded
// ------------------------------------------------------------------------------------------------
// PACKED: the same loop compiled for four workgroups per CU (128 VGPRs -- it spills --, half the segment table): the loop
// is latency-bound and co-resident workgroups overlap perfectly, so a batch of more than 2 x CUs signals finishes in
// fewer rounds; smaller batches run the roomier form.
template <typename R, bool PACKED = false> struct SparseRecorr {
    // LDS per workgroup is kept near 40 KB (4 workgroups per CU): the loop is latency bound, not compute bound
    static constexpr int kMaxSegments = PACKED ? 256 : 512;
    static constexpr bool kFused = false;
    static constexpr bool kLocomp = false;
    static constexpr int kMinWavesPerSimd = PACKED ? 4 : 1;
    static constexpr int kEnergyWaves = PACKED ? kWaves / 2 : kWaves;
    static constexpr bool kScoreOnly = false;
    using Shared = IterSharedT<R, kMaxSegments, false>;
    using Args = SparseArgs<R>;
    static constexpr int kGroup = 1;                    // one signal per workgroup: the hardware barrier
    using Sync = HwSync;
    static __device__ __forceinline__ Sync make_sync(Shared&) { return Sync(); }
    static __device__ __forceinline__ void prologue_shared(const DevParams&, const State<R>&, const Args&, char*) {}
    static __device__ __forceinline__ int signal_lds_offset(const DevParams&, const Args&) { return 0; }
    static __host__ __device__ bool has_bits(const DevParams& P, const Args& A) { return A.rl_cnt == nullptr && A.rowflag != nullptr && P.T <= kRowBitsMaxT; }
    // policy LDS: SparseLds | staged dictionary lists + weights (when small) | row bitmap (when T allows)
    static __host__ __device__ size_t bits_offset(const DevParams& P, const Args& A) { return sparse_lds_bytes<R>(A.caps) + staged_dict_bytes(P, A); }
    static size_t extra_lds_bytes(const DevParams& P, const Args& A)
    {
        return bits_offset(P, A) + (has_bits(P, A) ? (size_t)((P.T + 31) / 32) * sizeof(unsigned) : 0);
    }
    static __device__ __forceinline__ unsigned* bits_of(const DevParams& P, const Args& A, char* lds) { return reinterpret_cast<unsigned*>(lds + bits_offset(P, A)); }
    static __device__ __forceinline__ const R* weights(const DevParams& P, const State<R>&, const Args& A, char* lds)
    {
        return dict_view(P, A, lds + sparse_lds_bytes<R>(A.caps)).wts;
    }
    // the row-occupancy bitmap of this signal: flags written by the initial correlation (or an earlier launch)
    static __device__ __forceinline__ void prologue(const DevParams& P, const State<R>&, const Args& A, char* lds, int, Sync&)
    {
        stage_dict(P, A, lds + sparse_lds_bytes<R>(A.caps));
        if (!has_bits(P, A)) { __syncthreads(); return; }                  // (row lists need no per-launch state)
        unsigned* bits = bits_of(P, A, lds);
        const unsigned char* rf = A.rowflag + (int64_t)blockIdx.x * P.T;
        const int nwords = (P.T + 31) / 32;
        for (int i = threadIdx.x; i < nwords; i += kThreads) {
            unsigned word = 0;
            const int t0 = i * 32, n = min(32, P.T - t0);
            for (int q = 0; q < n; ++q) word |= rf[t0 + q] ? (1u << q) : 0u;
            bits[i] = word;
        }
        __syncthreads();
    }
    static __device__ __forceinline__ void epilogue(const DevParams& P, const State<R>&, const Args& A, char* lds, int)
    {
        if (!has_bits(P, A)) return;                        // (the caller's barrier made all bits visible)
        const unsigned* bits = bits_of(P, A, lds);
        unsigned char* rf = A.rowflag + (int64_t)blockIdx.x * P.T;
        for (int t = threadIdx.x; t < P.T; t += kThreads)
            if ((bits[t >> 5] >> (t & 31)) & 1u) rf[t] = 1;  // kept for resumed launches (hscmp_continue)
    }
    static __device__ __forceinline__ void resolve_wave(const DevParams&, const State<R>&, const Sig<R>&, const Args&, char*,
                                                        int, int, int&, R&) {}
    // The cells the selected atom touches may be non-zero after its subtraction: add them to their rows' lists.
    // Issued before the residual update so that its memory round trips overlap with it; the window gather of
    // run() comes several barriers later.  (Distinct (w, f) of one atom are distinct cells, so no two threads
    // append the same member.)
    // (the cells an atom touches are entered in the row lists by update_residual, next to its gather)
    static __device__ __forceinline__ void on_atom(const DevParams&, const State<R>&, const Args&, char*, int, int) {}
    // Residual update and window gather in one: away from the signal's ends the 3W-2 rows the re-correlation will gather
    // (run(), below) contain the atom's span, so they are gathered ONCE, before the subtraction; the atom's non-zeros
    // are applied to the LDS copy as well as to the residual, the energy partial sums are taken over the span's part,
    // and the list (L.key / L.val in gather_window's encoding, count in ctl[0]) is left for sparse_rows (ctl[3] = -2; other values of
    // that word are list lengths and flags >= 0).
    // Returns false, with nothing changed but the row lists, when a row list of the window has overflowed, the cells
    // do not fit, or three cells meet in one partial sum: update_residual then runs its span-only form.
    static __device__ __forceinline__ bool merged_update(const DevParams& P, const Sig<R>& G, const Args& A, const SparseLds<R>& L,
                                                         int p, int k, R c, int s, int e, R& pb, R& pa)
    {
        const int T = P.T, F = P.F, W = P.W, tid = laundered_tid<PACKED>();
        const int g0 = p - P.off - (W - 1), nwin = 3 * W - 2;
        const int* cnt = A.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A.rl_f + (int64_t)blockIdx.x * T * 8;
        int* cntw = A.rl_cnt + (int64_t)blockIdx.x * T;
        int* lfw = A.rl_f + (int64_t)blockIdx.x * T * 8;
        int* key = L.key; R* cur = L.val; R* before = L.rx;
        int* members = L.perm; int* slot0 = reinterpret_cast<int*>(L.okey); int* slot1 = reinterpret_cast<int*>(L.rkey);   // [256] each
        const int e0 = A.nzptr[k], e1 = A.nzptr[k + 1], na = e1 - e0;
        if (tid == 0) { L.ctl[0] = 0; L.ctl[1] = 0; L.ctl[2] = 0; L.ctl[3] = 0; }
        members[tid] = 0;
        HSCMP_STAMP_BEGIN();
        lds_barrier();
        HSCMP_STAMP(52);
        // (a) row threads, from thread 0 up: count + list of a window row, then its cells; (b) atom threads, from the last
        // thread down: the atom's non-zero, then the list of its row, and an unlisted cell takes the next free slot
        // (the first 256 non-zeros of the atom stay with their threads: value and list slot are used further down)
        int my_wf = 0, pend_slot = 8, pend_row = -1;
        R my_val = (R)0;
        auto flush = [&]() { if (pend_row >= 0 && pend_slot < 8) lfw[(int64_t)pend_row * 8 + pend_slot] = my_wf & 0xffff; };
        for (int base = 0; base < max(nwin, na); base += kThreads) {
            const int j = base + tid;
            const bool row_on = j < nwin;
            const int g = g0 + (row_on ? j : 0);
            const int qa = base + (kThreads - 1 - tid);
            const bool atom_on = qa < na;
            const int q = e0 + (atom_on ? qa : 0);
            const int n = list_count(cnt + g);
            const int4* row = reinterpret_cast<const int4*>(lf + (int64_t)g * 8);
            const int4 a = row[0], b = row[1];
            const int wf = na > 0 ? A.nzwf[q] : 0;
            if (base == 0 && atom_on) { my_wf = wf; my_val = A.nzval[q]; }
            const int fa = wf & 0xffff;
            const int ga = atom_on ? p - P.off + (wf >> 16) : g0;         // inside the window, which is inside the signal
            const int fs[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const bool cells_on = row_on && n > 0 && n <= 8;
            R vs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vs[u] = (cells_on && fs[u] >= 0) ? G.r[(int64_t)g * F + fs[u]] : (R)0;
            const int n2 = list_count(cnt + ga);
            const int4* row2 = reinterpret_cast<const int4*>(lf + (int64_t)ga * 8);
            const int4 a2 = row2[0], b2 = row2[1];
            if (row_on && n > 8) atomicAdd(&L.ctl[1], 1);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (!(vs[u] != (R)0)) continue;
                const int o = atomicAdd(&L.ctl[0], 1);
                if (o < L.caps.nz) { key[o] = (fs[u] << 16) | j; before[o] = vs[u]; cur[o] = vs[u]; }
            }
            if (atom_on) {
                const bool listed = n2 > 8 || a2.x == fa || a2.y == fa || a2.z == fa || a2.w == fa ||
                                    b2.x == fa || b2.y == fa || b2.z == fa || b2.w == fa;
                if (!listed) {
                    const int o = atomicAdd(&cntw[ga], 1);
                    // (the slot number is only needed for the store of the feature index: that store waits until the end
                    //  of this function, so nobody waits for the atomic's round trip here)
                    if (base == 0) { pend_slot = o; pend_row = ga; }
                    else if (o < 8) lfw[(int64_t)ga * 8 + o] = fa;
                }
            }
        }
        HSCMP_STAMP(53);
        lds_barrier();
        HSCMP_STAMP(54);
        const int n0 = L.ctl[0];
        if (L.ctl[1] > 0 || n0 + na > L.caps.nz || na > kThreads) { flush(); return false; }      // uniform
        // Partial sums first (nothing is written to the residual before the form is settled): the span's cells register
        // with partial sum (cell index mod 256); an atom cell that is not in the list yet joins with value 0.
        const int js = s - g0, je = e - g0;                                           // the span's rows inside the window
        const R nc = -c;
        int slot = -1;                                                              // atom thread: list entry of its cell
        R prod = (R)0;
        int wfq = 0;
        {
            const int qa = kThreads - 1 - tid;
            if (qa < na) {                                                           // (na <= 256: one per thread)
                wfq = my_wf;
                prod = nc * my_val;                                                  // -c*D[k] rounded, then += (utils.py:120,129)
                const int kk = ((wfq & 0xffff) << 16) | (p - P.off + (wfq >> 16) - g0);
                int jj = n0;
                for (int j0 = 0; j0 < n0 && jj == n0; j0 += 4) {                    // four list entries per LDS round trip
                    const int k0 = key[j0], k1 = j0 + 1 < n0 ? key[j0 + 1] : -1, k2 = j0 + 2 < n0 ? key[j0 + 2] : -1, k3 = j0 + 3 < n0 ? key[j0 + 3] : -1;
                    jj = k0 == kk ? j0 : k1 == kk ? j0 + 1 : k2 == kk ? j0 + 2 : k3 == kk ? j0 + 3 : n0;
                }
                if (jj < n0) slot = jj;
                else { slot = n0 + atomicAdd(&L.ctl[2], 1); key[slot] = kk; before[slot] = (R)0; cur[slot] = (R)0; }   // (ctl[0] is still being read)
            }
        }
        lds_barrier();
        HSCMP_STAMP(55);
        const int n = n0 + L.ctl[2];
        for (int i = tid; i < n; i += kThreads) {
            const int kk = key[i], j = kk & 0xffff;
            if (j < js || j >= je) continue;
            const int cq = ((j - js) * F + (kk >> 16)) & (kThreads - 1);
            const int at = atomicAdd(&members[cq], 1);
            if (at == 0) slot0[cq] = i; else if (at == 1) slot1[cq] = i; else L.ctl[3] = 1;
        }
        lds_barrier();
        HSCMP_STAMP(56);
        if (L.ctl[3] != 0) { flush(); return false; }                                 // uniform; the list appends stay (harmless)
        // the atom's non-zeros: residual and LDS copy
        if (slot >= 0) {
            const R vn = before[slot] + prod;
            cur[slot] = vn;
            G.r[(int64_t)(p - P.off + (wfq >> 16)) * F + (wfq & 0xffff)] = vn;
        }
        lds_barrier();
        const int m = members[tid];
        if (m > 0) {
            const int i = slot0[tid];
            const R b = before[i], a = cur[i];
            const R sb = b * b, sa = a * a;
            pb = pb + sb; pa = pa + sa;
        }
        if (m > 1) {
            const int i = slot1[tid];
            const R b = before[i], a = cur[i];
            const R sb = b * b, sa = a * a;
            pb = pb + sb; pa = pa + sa;
        }
        flush();
        HSCMP_STAMP(57);
        if (tid == 0) { L.ctl[0] = n; L.ctl[3] = -2; }                                 // the window list is ready for run()
        return true;
    }
    // Residual subtraction with the local energies (modeling.py:996-1016) over the listed cells only.  The dense form
    // streams the whole W x F window and the dense atom (96 KB per atom at BASELINE config 4: the loop's HBM traffic);
    // here the span's listed non-zero cells are gathered, the atom's few non-zeros are applied to them (new cells
    // start from 0), and the pinned energy order -- partial sum (i mod 256), each sequential in the cell index i --
    // is kept by summing the cells in sorted order.  A zero cell adds +0 to its partial sum and an atom zero changes
    // nothing, so the results are those of the dense form.  Returns false (nothing done) when a row list of the span
    // has overflowed or the cells do not fit the LDS list: the dense form then runs.
    static __device__ __forceinline__ bool update_residual(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A0, char* lds,
                                                           int p, int k, R c, int s, int e, int es, R& pb, R& pa)
    {
        if (!A0.rl_cnt) return false;
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        const SparseLds<R> L = sparse_lds_view<R>(lds, A0.caps);
        const int T = P.T, F = P.F, tid = laundered_tid<PACKED>(), C = A.rl_cap, shift = __ffs(C) - 1;
        const int* cnt = A.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A.rl_f + (int64_t)blockIdx.x * T * C;
        {
            const int g0 = p - P.off - (P.W - 1);
            if (C == 8 && g0 >= 0 && g0 + 3 * P.W - 2 <= T && 3 * P.W - 2 <= 0xffff) {
                if (merged_update(P, G, A, L, p, k, c, s, e, pb, pa)) return true;
                lds_barrier();               // every thread has read the counters that settled it before they are reset below
            }
        }
        int* key = L.key; R* before = L.val; R* after = L.rd; int* order = L.perm;
        const int e0 = A.nzptr[k], e1 = A.nzptr[k + 1];
        int* members = L.perm; int* slot0 = reinterpret_cast<int*>(L.okey); int* slot1 = reinterpret_cast<int*>(L.rkey);   // [256] each
        if (tid == 0) { L.ctl[0] = 0; L.ctl[1] = 0; L.ctl[2] = 0; L.ctl[3] = 0; }
        members[tid] = 0;
        lds_barrier();                                                                // (LDS lists only; the global stores drain at the caller's barrier)
        // Two jobs share the memory round trips.  (a) Row threads (from thread 0 up) gather the span's listed cells:
        // a row's count and list in one trip, its cells in the next.  (b) Atom threads (from the last thread down)
        // enter the atom's cells in the row lists (the window gathers read them instead of scanning rows of F values):
        // the atom's non-zero, then the row's list; an unlisted cell takes the next free slot.  A cell entered under (b)
        // still holds 0 and is skipped by (a) whether (a) sees it or not; a count pushed past the capacity is seen by
        // every later reader as "read the row densely".
        const int na = e1 - e0;
        int* cntw = A.rl_cnt + (int64_t)blockIdx.x * T;
        int* lfw = A.rl_f + (int64_t)blockIdx.x * T * C;
        if (C == 8) {
            for (int base = 0; base < max(e - s, na); base += kThreads) {
                const int g = s + base + tid;                                // (a) this thread's row
                const bool row_on = g < e;
                const int gq = row_on ? g : s;
                const int qa = base + (kThreads - 1 - tid);                  // (b) this thread's non-zero of the atom
                const bool atom_on = qa < na;
                const int q = e0 + (atom_on ? qa : 0);
                // first trip
                const int n = list_count(cnt + gq);
                const int4* row = reinterpret_cast<const int4*>(lf + (int64_t)gq * 8);
                const int4 a = row[0], b = row[1];
                const int wf = na > 0 ? A.nzwf[q] : 0;
                // second trip
                const int fa = wf & 0xffff;
                int ga = p - P.off + (wf >> 16);
                const bool cell_on = atom_on && ga >= 0 && ga < T;            // clipped part of the atom (utils.py:110-129)
                if (!cell_on) ga = s;
                const int fs[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                const bool cells_on = row_on && n > 0 && n <= C;
                R vs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vs[u] = (cells_on && fs[u] >= 0) ? G.r[(int64_t)g * F + fs[u]] : (R)0;
                const int n2 = list_count(cnt + ga);
                const int4* row2 = reinterpret_cast<const int4*>(lf + (int64_t)ga * 8);
                const int4 a2 = row2[0], b2 = row2[1];
                if (row_on && n > C) atomicAdd(&L.ctl[1], 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (!(vs[u] != (R)0)) continue;
                    const int o = atomicAdd(&L.ctl[0], 1);
                    if (o < L.caps.nz) { key[o] = (g - s) * F + fs[u]; before[o] = vs[u]; after[o] = vs[u]; }
                }
                if (cell_on) {
                    // (empty slots hold -1 and never match; an overflowed row, count > C, is read densely anyway)
                    const bool listed = n2 > C || a2.x == fa || a2.y == fa || a2.z == fa || a2.w == fa ||
                                        b2.x == fa || b2.y == fa || b2.z == fa || b2.w == fa;
                    if (!listed) {
                        const int o = atomicAdd(&cntw[ga], 1);
                        if (o < C) lfw[(int64_t)ga * C + o] = fa;
                    }
                }
            }
        } else {
            for (int q = e0 + tid; q < e1; q += kThreads) {
                const int wf = A.nzwf[q], f = wf & 0xffff, g = p - P.off + (wf >> 16);
                if (g < 0 || g >= T) continue;
                bool listed = list_count(cnt + g) > C;
                for (int u = 0; u < C; ++u) listed |= lf[(int64_t)g * C + u] == f;
                if (!listed) {
                    const int o = atomicAdd(&cntw[g], 1);
                    if (o < C) lfw[(int64_t)g * C + o] = f;
                }
            }
            const int items = (e - s) << shift;
            for (int it = tid; it < items; it += kThreads) {
                const int g = s + (it >> shift);
                const int n = list_count(cnt + g);
                if (n > C) { if ((it & (C - 1)) == 0) atomicAdd(&L.ctl[1], 1); continue; }
                const int f = lf[((int64_t)g << shift) + (it & (C - 1))];
                if (f < 0) continue;
                const R v = G.r[(int64_t)g * F + f];
                if (v != (R)0) {
                    const int o = atomicAdd(&L.ctl[0], 1);
                    if (o < L.caps.nz) { key[o] = (g - s) * F + f; before[o] = v; after[o] = v; }
                }
            }
        }
        lds_barrier();                                                                // (LDS lists only; the global stores drain at the caller's barrier)
        const int n0 = L.ctl[0];
        if (L.ctl[1] > 0 || n0 + (e1 - e0) > L.caps.nz) return false;                 // uniform
        // the atom's non-zeros: -c*D[k] rounded, then += (utils.py:120,129)
        const R nc = -c;
        for (int q = e0 + tid; q < e1; q += kThreads) {
            const int wf = A.nzwf[q], f = wf & 0xffff, g = p - P.off + (wf >> 16);
            if (g < s || g >= e) continue;                                           // clipped part of the atom
            const int i = (g - s) * F + f;
            const R prod = nc * A.nzval[q];
            int j = 0;
            while (j < n0 && key[j] != i) ++j;
            R vn;
            if (j < n0) { vn = before[j] + prod; after[j] = vn; }
            else {
                vn = (R)0 + prod;
                const int o = n0 + atomicAdd(&L.ctl[2], 1);                           // (ctl[0] is still being read)
                key[o] = i; before[o] = (R)0; after[o] = vn;
            }
            G.r[(int64_t)g * F + f] = vn;
        }
        lds_barrier();                                                                // (LDS lists only; the global stores drain at the caller's barrier)
        const int n = n0 + L.ctl[2];
        // Energy partial sums: partial q sums the cells with index = q mod 256 in ascending order.  With at most two
        // cells per partial sum the order does not matter (0 + a = a, a + b = b + a): every cell registers with its
        // partial sum and thread q adds what it finds.  A third cell somewhere: sort and walk, as the order prescribes.
        for (int j = tid; j < n; j += kThreads) {
            const int cq = key[j] & (kThreads - 1);
            const int at = atomicAdd(&members[cq], 1);
            if (at == 0) slot0[cq] = j; else if (at == 1) slot1[cq] = j; else L.ctl[3] = 1;
        }
        lds_barrier();                                                                // (LDS lists only; the global stores drain at the caller's barrier)
        if (L.ctl[3] == 0) {                                                          // uniform
            const int m = members[tid];
            if (m > 0) {
                const int j = slot0[tid];
                const R b = before[j], a = after[j];
                const R sb = b * b, sa = a * a;
                pb = pb + sb; pa = pa + sa;
            }
            if (m > 1) {
                const int j = slot1[tid];
                const R b = before[j], a = after[j];
                const R sb = b * b, sa = a * a;
                pb = pb + sb; pa = pa + sa;
            }
            return true;
        }
        for (int j = tid; j < n; j += kThreads) {                                     // rank sort by cell index (distinct)
            const int kj = key[j];
            int rank = 0;
            for (int q = 0; q < n; ++q) rank += (key[q] < kj) ? 1 : 0;
            order[rank] = j;
        }
        __syncthreads();
        for (int q = 0; q < n; ++q) {                                                 // partial sum tid, ascending cell index
            const int j = order[q];
            if ((key[j] & (kThreads - 1)) == tid) {
                const R b = before[j], a = after[j];
                const R sb = b * b, sa = a * a;
                pb = pb + sb; pa = pa + sa;
            }
        }
        return true;                                                                  // (pinned_tree2 of the caller synchronises)
    }
    // Energy partial sums of the clipped window [s, e) x F over its listed cells (weak-atom filter, :1090-1099), in the
    // pinned order like update_residual.  Returns false when a row list of the window has overflowed or the cells
    // do not fit: the dense form then runs.  All threads call it (barriers); p is this thread's partial sum.
    static __device__ __forceinline__ bool window_partials(const DevParams& P, const Sig<R>& G, const Args& A0, char* lds, int s, int e, R& p)
    {
        if (!A0.rl_cnt) return false;
        const SparseLds<R> L = sparse_lds_view<R>(lds, A0.caps);
        const int T = P.T, F = P.F, tid = laundered_tid<PACKED>(), C = A0.rl_cap, shift = __ffs(C) - 1;
        const int* cnt = A0.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A0.rl_f + (int64_t)blockIdx.x * T * C;
        int* key = L.key; R* val = L.val; int* order = L.perm;
        __syncthreads();                                                             // the lists of the previous window are consumed
        if (tid == 0) { L.ctl[0] = 0; L.ctl[1] = 0; }
        __syncthreads();
        const int items = (e - s) << shift;
        for (int it = tid; it < items; it += kThreads) {
            const int g = s + (it >> shift);
            const int n = list_count(cnt + g);
            if (n > C) { if ((it & (C - 1)) == 0) atomicAdd(&L.ctl[1], 1); continue; }
            const int f = lf[((int64_t)g << shift) + (it & (C - 1))];
            if (f < 0) continue;
            const R v = G.r[(int64_t)g * F + f];
            if (v != (R)0) {
                const int o = atomicAdd(&L.ctl[0], 1);
                if (o < L.caps.nz) { key[o] = (g - s) * F + f; val[o] = v; }
            }
        }
        __syncthreads();
        const int n = L.ctl[0];
        if (L.ctl[1] > 0 || n > L.caps.nz) return false;                             // uniform
        for (int j = tid; j < n; j += kThreads) {
            const int kj = key[j];
            int rank = 0;
            for (int q = 0; q < n; ++q) rank += (key[q] < kj) ? 1 : 0;
            order[rank] = j;
        }
        __syncthreads();
        for (int q = 0; q < n; ++q) {
            const int j = order[q];
            if ((key[j] & (kThreads - 1)) == tid) { const R sq = val[j] * val[j]; p = p + sq; }
        }
        return true;
    }
    // The same energy by ONE wave, without workgroup barriers (the weak-atom filter gives every wave its own
    // candidates): lane = row of the window; its listed cells go to the slot of their partial sum i mod 256 in the
    // wave's LDS table.  A partial sum with one or two cells is order free (0 + a = a, a + b = b + a); a window with
    // three cells in one partial sum, or with an overflowed row list, is left to window_partials (returns false,
    // wave-uniform).  The tree is that of wave_window_energy.  Result in lane 0.
    static __device__ __forceinline__ bool wave_window_listed(const DevParams& P, const Sig<R>& G, const Args& A0, char* lds,
                                                              int s, int e, int lane, int wv, R& out)
    {
        if (!A0.rl_cnt || A0.rl_cap != 8) return false;
        const int T = P.T, F = P.F;
        const int* cnt = A0.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A0.rl_f + (int64_t)blockIdx.x * T * 8;
        // (wv < kEnergyWaves: the caller lets the waves in by turns when the table holds fewer than four windows)
        R* first = reinterpret_cast<R*>(lds) + (size_t)wv * kThreads;                       // [kEnergyWaves][256]
        R* second = reinterpret_cast<R*>(lds) + (size_t)(kEnergyWaves + wv) * kThreads;     // [kEnergyWaves][256]
        int* members = reinterpret_cast<int*>(lds + (size_t)2 * kEnergyWaves * kThreads * sizeof(R)) + (size_t)wv * kThreads;
#pragma unroll
        for (int c = 0; c < 4; ++c) members[lane + 64 * c] = 0;
        bool bad = false;                                   // (one wave: its LDS operations execute in order)
        for (int t0 = s; t0 < e; t0 += 64) {
            const int g = t0 + lane;
            const int n = g < e ? list_count(cnt + g) : 0;
            if (n > 8) bad = true;
            if (n > 0 && n <= 8) {
                const int4* row = reinterpret_cast<const int4*>(lf + (int64_t)g * 8);
                const int4 a = row[0], b = row[1];
                const int fs[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                R vs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vs[u] = fs[u] >= 0 ? G.r[(int64_t)g * F + fs[u]] : (R)0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (fs[u] < 0) continue;
                    const int c = ((g - s) * F + fs[u]) & (kThreads - 1);
                    const int at = atomicAdd(&members[c], 1);
                    const R sq = vs[u] * vs[u];
                    if (at == 0) first[c] = sq; else if (at == 1) second[c] = sq;
                }
            }
        }
        R p[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int m = members[lane + 64 * c];
            if (m > 2) bad = true;
            const R x0 = m > 0 ? first[lane + 64 * c] : (R)0, x1 = m > 1 ? second[lane + 64 * c] : (R)0;
            const R h = (R)0 + x0;
            p[c] = h + x1;
        }
        if (__ballot(bad) != 0ull) return false;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const R o = __shfl_down(p[c], m); p[c] = p[c] + o; }
        }
        const R a01 = p[0] + p[1], a23 = p[2] + p[3];
        out = a01 + a23;
        return true;
    }
    // After update_residual(): true when merged_update left the updated window in LDS for run() (uniform: the flag is
    // written before the barriers of the energy tree)
    static __device__ __forceinline__ bool residual_copy_in_lds(const Args& A0, char* lds)
    {
        return A0.rl_cnt && sparse_lds_view<R>(lds, A0.caps).ctl[3] == -2;
    }
    // After run(): the rows p-(W-1) .. p+(W-1) as the arg-max over listed outputs left them in LDS (see sparse_rows, step d)
    static __device__ __forceinline__ bool row_results(const DevParams& P, const Args& A0, char* lds, int p, const int*& rk, const R*& rc,
                                                       const R*& r0, int& t0, int& n)
    {
        const SparseLds<R> L = sparse_lds_view<R>(lds, A0.caps);
        if (L.ctl[3] != -3) return false;                  // uniform: written before the last barrier of sparse_rows
        rk = L.perm; rc = L.rd; r0 = L.rx; t0 = p - (P.W - 1); n = 2 * P.W - 1;
        return true;
    }
    template <typename SH>
    static __device__ __forceinline__ void run(const DevParams& P, const State<R>& S, const Sig<R>& G, SH&, const Args& A0,
                                               char* lds, int p, int k)
    {
        HSCMP_STAMP_BEGIN();
        const SparseLds<R> L = sparse_lds_view<R>(lds, A0.caps);
        const int T = P.T, W = P.W;
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        unsigned* bits = has_bits(P, A) ? bits_of(P, A0, lds) : nullptr;
        if (A.rl_cnt) {
            // (the atom's cells were entered in the row lists by update_residual)
        } else if (bits) {
            // the atom just subtracted made its span possibly non-zero (utils.py:76-131)
            int s, e, es;
            centered_span(T, W, p, s, e, es);
            for (int t = s + (int)threadIdx.x; t < e; t += kThreads) atomicOr(&bits[t >> 5], 1u << (t & 31));
        }                                                  // (ordered by the first barrier of gather_window)
        const int tstart = p - P.off - (W - 1);            // :1028-1033
        const int tend = p + W / 2 + (W - 1);              // :1038
        const int sidx = tstart < 0 ? 0 : tstart;          // :1034
        const int eidx = tend > T - 1 ? T - 1 : tend;      // :1039
        HSCMP_STAMP(45);
        const bool gathered = A.rl_cnt && L.ctl[3] == -2;     // uniform: written before the barriers of the energy tree
        sparse_rows<R, PACKED>(P, S, G, A, L, bits, p - (W - 1), 2 * W - 1, true, sidx, eidx - sidx + 1, gathered);
    }
};

// ------------------------------------------------------------------------------------------------
// initial correlation of a sparse multi-feature input (modeling.py:1077): only rows whose window
// contains a non-zero input row can be non-zero; everything else is (c = 0, k = 0).
//   grid = (B, nsplit): workgroup (b, y) owns a contiguous range of (2W-1)-row blocks of signal b;
//   block = kThreads;  dynamic LDS = SparseLds<R> + staged dictionary lists + T bits of row flags.
//   A.scratch needs B * nsplit tables.
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(kThreads) void corr_init_sparse_kernel(DevParams P, State<R> S, SparseArgs<R> A0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SparseLds<R> L = sparse_lds_view<R>(smem, A0.caps);
    unsigned* rowbits = reinterpret_cast<unsigned*>(smem + sparse_lds_bytes<R>(A0.caps) + staged_dict_bytes(P, A0));   // [ceil(T/32)]
    const int b = blockIdx.x, tid = threadIdx.x, T = P.T, W = P.W, F = P.F;
    stage_dict(P, A0, smem + sparse_lds_bytes<R>(A0.caps));               // (ordered by the barrier below)
    SparseArgs<R> A = dict_view(P, A0, smem + sparse_lds_bytes<R>(A0.caps));
    A.scratch = A0.scratch + ((int64_t)blockIdx.y * gridDim.x) * (2 * W - 1) * P.K;   // sparse_rows adds blockIdx.x tables
    Sig<R> G{};
    G.r = S.residual + (int64_t)b * T * F;
    G.bc = S.best_c + (int64_t)b * T;
    G.bk = S.best_k + (int64_t)b * T;
    // this workgroup's output rows [r_lo, r_hi) and the input rows they can see [in_lo, in_hi)
    const int blk = 2 * W - 1;
    const int nblocks = (T + blk - 1) / blk;
    const int per = (nblocks + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_lo = min(T, (int)blockIdx.y * per * blk), r_hi = min(T, ((int)blockIdx.y + 1) * per * blk);
    if (r_lo >= r_hi) return;
    const int in_lo = max(0, r_lo - P.off), in_hi = min(T, r_hi - P.off + W - 1);
    for (int i = (in_lo >> 5) + tid; i <= ((in_hi - 1) >> 5); i += kThreads) rowbits[i] = 0u;
    for (int t = r_lo + tid; t < r_hi; t += kThreads) { G.bc[t] = (R)0; G.bk[t] = 0; }
    __syncthreads();
    // input rows with a non-zero sample: handed over by the level chaining, or found by a scan
    unsigned char* rf = A.rowflag ? A.rowflag + (int64_t)b * T : nullptr;
    if (A.rl_cnt) {
        const int* cnt = A.rl_cnt + (int64_t)b * T;
        for (int t = in_lo + tid; t < in_hi; t += kThreads) if (cnt[t] > 0) atomicOr(&rowbits[t >> 5], 1u << (t & 31));
    } else if (rf && A.rowflag_filled) {
        for (int t = in_lo + tid; t < in_hi; t += kThreads) if (rf[t]) atomicOr(&rowbits[t >> 5], 1u << (t & 31));
    } else {
        const int64_t e0 = (int64_t)in_lo * F, e1 = (int64_t)in_hi * F;
        for (int64_t e = e0 + tid; e < e1; e += kThreads)
            if (G.r[e] != (R)0) atomicOr(&rowbits[(int)(e / F) >> 5], 1u << ((int)(e / F) & 31));
    }
    __syncthreads();
    if (rf && !A.rowflag_filled && !A.rl_cnt)            // publish the flags of the owned rows for the greedy loop
        for (int t = r_lo + tid; t < r_hi; t += kThreads) rf[t] = (unsigned char)((rowbits[t >> 5] >> (t & 31)) & 1u);
    // walk the output rows in blocks of up to 2W-1; a block is computed iff some input row in its reach is set
    for (int row0 = r_lo; row0 < r_hi; row0 += blk) {
        const int nrows = min(blk, T - row0);
        const int lo = max(0, row0 - P.off), hi = min(T - 1, row0 + nrows - 1 - P.off + W - 1);
        int any = 0;
        for (int t = lo + tid; t <= hi; t += kThreads) any |= (rowbits[t >> 5] >> (t & 31)) & 1u;
        if (__syncthreads_or(any)) sparse_rows(P, S, G, A, L, rowbits, row0, nrows, false, 0, 0);
    }
}

}  // namespace hscmp
