// LoCOMP inside the greedy loop: the joint re-fit of a selected atom and its neighbourhood (hsc/modeling.py:1191-1425;
// Mailhe et al., ICASSP 2009) as the atom body of iterate_kernel.
//
// The reference's loop differs from ConvolutionalMatchingPursuit's (:1053-1186) in what happens to a selected atom
// (p, k, c):
//   :1222-1241  the previously selected atoms whose position lies within [start - W/2, end + W/2 (- 1)] of the atom's span
//               -- minus those that share its dictionary index, minus those whose row INSIDE THAT WINDOW equals the atom's
//               absolute position (the reference compares the two as they are; reproduced) -- form its group;
//   :1322-1341  the coefficients of the group are re-fitted on the local residual: least squares over the union of the
//               supports (np.linalg.pinv there; here the normal equations of the same system in float64, Cholesky -- the
//               Gram matrix of a handful of atoms), ADDED to the stored coefficients, and every atom of the group is
//               removed from the residual and re-correlated around (:1343-1353);
//   :1368-1383  `coefficients.nnz` (stored non-zeros: an entry that cancels to 0.0 leaves the count) is what nbNonzeroCoefs
//               is compared with, and the loop also stops when an atom changes the residual energy by less than eps.
// Selection, weak-atom filter, residual subtraction with its local energies, local re-correlation and the round-level stop
// rules are those of the greedy loop and run through the same code (GenericRecorr: the table-free dense form).
// Parity is at tolerance level by construction (the reference's pseudo-inverse is an SVD in the dictionary's dtype).
#pragma once
#include "hscmp_kernels.h"
#include "hscmp_sparse.h"

namespace hscmp {

constexpr int kLocompMax = 64;        // atoms of a group, the selected one included (larger: STOP_GROUP, the host loop takes over)
// Nearly dependent atoms in a group (a Cholesky pivot that all but vanishes against its diagonal entry): the re-fit is solved in
// float64 whatever the dictionary's dtype, an atom the others already span keeps coefficient 0.  The reference's pseudo-inverse is
// an SVD in the DICTIONARY's dtype with a cut-off of 1e-15: on such a group its float32 result is round-off amplified by the
// condition number, which no other solver reproduces -- there the two agree on the residual they leave, not coefficient by
// coefficient (the hierarchical per-signal entry therefore keeps the reference's own LAPACK call on the host).
constexpr double kLocompDead = 1e-12;

template <typename R> struct LocompLds {
    int n, cnt;                       // group size; neighbours found (may exceed the capacity)
    int t[kLocompMax], k[kLocompMax], si[kLocompMax];       // position, atom, coefficient slot (-1: none yet), group order
    int ut[kLocompMax], uk[kLocompMax], usi[kLocompMax];    // neighbours as found (any order)
    R a[kLocompMax];                  // fitted coefficients in the dictionary's dtype (:1329)
    R loss, last_e;                   // energyLoss of the group (:998-1014), lastEnergyResidual (:1316)
    double b[kLocompMax];             // right-hand side <d_i, r>, then the solution
    double diag[kLocompMax];          // original diagonal (rank test)
    double g[kLocompMax * kLocompMax];    // Gram matrix <d_i, d_j> of the clipped atoms, then its Cholesky factor (lower)
};

// the dense table-free loop (GenericRecorr) with the group re-fit as its atom body
template <typename R> struct LocompRecorr : GenericRecorr<R> {
    static constexpr bool kLocomp = true;
    using Base = GenericRecorr<R>;
    using Args = typename Base::Args;
    static size_t extra_lds_bytes(const DevParams& P) { return Base::extra_lds_bytes(P) + sizeof(LocompLds<R>) + 16; }
    static __device__ __forceinline__ LocompLds<R>& group(const DevParams&, const Args&, char* lds)
    {
        return *reinterpret_cast<LocompLds<R>*>(lds + ((Base::kWinBytes + 15) / 16) * 16);
    }
    static __device__ __forceinline__ void before_runs(const Args&, char*) {}
};

// the same on the sparse policy (multi-feature inputs, sparse dictionary: hierarchical levels >= 1): the residual update keeps
// the per-row lists of non-zero cells current, the re-correlation forms the non-zero products only
template <typename R> struct LocompSparse : SparseRecorr<R, false> {
    static constexpr bool kLocomp = true;
    using Base = SparseRecorr<R, false>;
    using Args = typename Base::Args;
    static size_t policy_bytes(const DevParams& P, const Args& A) { return ((Base::extra_lds_bytes(P, A) + 15) / 16) * 16; }
    static size_t extra_lds_bytes(const DevParams& P, const Args& A) { return policy_bytes(P, A) + sizeof(LocompLds<R>) + 16; }
    static __device__ __forceinline__ LocompLds<R>& group(const DevParams& P, const Args& A, char* lds)
    {
        const size_t off = ((Base::bits_offset(P, A) + (Base::has_bits(P, A) ? (size_t)((P.T + 31) / 32) * sizeof(unsigned) : 0) + 15) / 16) * 16;
        return *reinterpret_cast<LocompLds<R>*>(lds + off);
    }
    // the group's subtractions all come before its re-correlations: the window copy the last subtraction left in LDS for "its"
    // re-correlation (SparseRecorr::merged_update) belongs to no atom that follows
    static __device__ __forceinline__ void before_runs(const Args& A, char* lds)
    {
        if (ltid() == 0) sparse_lds_view<R>(lds, A.caps).ctl[3] = 0;
    }
};

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// One selected atom (p, k, c): modeling.py:1314-1383.  All threads of the signal's workgroup; the caller leaves the atom
// loop when sh.skip or sh.converged is set afterwards.
template <typename R, typename Pol, typename SH, typename SY>
__device__ __forceinline__ void locomp_atom(const DevParams& P, const State<R>& S, const Sig<R>& G, SH& sh, const typename Pol::Args& A,
                                            char* plds, const R* wts, int p, int k, R c, SY& sy)
{
    const int T = P.T, W = P.W, F = P.F, tid = ltid(), lane = tid & 63, wv = tid >> 6;
    LocompLds<R>& L = Pol::group(P, A, plds);
    // ---- event list, the atom's own entry, its neighbourhood (:1222-1241)
    if (tid == 0) {
        if (sh.nev >= P.cap) { sh.converged = 1; sh.stop = STOP_CAPACITY; sh.skip = 1; }
        L.cnt = 0; L.t[0] = p; L.k[0] = k; L.si[0] = -1; L.loss = (R)0; L.last_e = sh.e_res;
    }
    sy.full();
    if (sh.skip) return;
    int s0, e0, es0;
    centered_span(T, W, p, s0, e0, es0);
    const int nstart = max(s0 - W / 2, 0);
    const int nend = min((e0 - 1) + ((W & 1) ? W / 2 : W / 2 - 1), T);
    const int ns = sh.nslots;
    for (int i = tid; i < ns; i += kThreads) {
        const int ti = G.slot_t[i], ki = G.slot_k[i];
        if (ti == p && ki == k) L.si[0] = i;                     // (at most one)
        if (ti < nstart || ti > nend || ki == k || (ti - nstart) == p) continue;
        if (!(G.slot_a[i] != 0.0)) continue;                      // (the list-of-lists matrix drops an entry that became 0.0)
        const int o = atomicAdd(&L.cnt, 1);
        if (o < kLocompMax - 1) { L.ut[o] = ti; L.uk[o] = ki; L.usi[o] = i; }
    }
    sy.full();
    const int m = L.cnt;
    if (m > kLocompMax - 1) {                                    // uniform
        if (tid == 0) { sh.converged = 1; sh.stop = STOP_GROUP; sh.skip = 1; }
        sy.full();
        return;
    }
    // group order: the new atom, then the neighbours by (position, atom) -- the order of the reference's sparse slice
    if (tid < m) {
        const long long key = ((long long)L.ut[tid] << 32) | (unsigned)L.uk[tid];
        int rank = 0;
        for (int q = 0; q < m; ++q) rank += ((((long long)L.ut[q] << 32) | (unsigned)L.uk[q]) < key) ? 1 : 0;
        L.t[1 + rank] = L.ut[tid]; L.k[1 + rank] = L.uk[tid]; L.si[1 + rank] = L.usi[tid];
    }
    if (tid == 0) { L.n = 1 + m; L.a[0] = c; }
    sy.full();
    const int n = 1 + m;

    if (n > 1) {
        // ---- :1322-1329 least squares of the local residual on the group's (clipped) atoms: G x = b, float64
        // items: n right-hand sides, then the n (n + 1) / 2 Gram entries; one wave per item, lanes over the elements
        const int nitems = n + n * (n + 1) / 2;
        for (int it = wv; it < nitems; it += kWaves) {
            double acc = 0.0;
            if (it < n) {
                int s, e, es;
                const int len = centered_span(T, W, L.t[it], s, e, es);
                const R* dk = S.D + ((int64_t)L.k[it] * W + es) * F;
                const R* rv = G.r + (int64_t)s * F;
                for (int i = lane; i < len * F; i += 64) acc += (double)dk[i] * (double)rv[i];
                acc = wave_sum_f64(acc);
                if (lane == 0) L.b[it] = acc;
            } else {
                // (row i, column j <= i) from the linear index of the lower triangle
                int q = it - n, i = 0;
                while ((i + 1) * (i + 2) / 2 <= q) ++i;
                const int j = q - i * (i + 1) / 2;
                int si_, ei_, esi, sj_, ej_, esj;
                centered_span(T, W, L.t[i], si_, ei_, esi);
                centered_span(T, W, L.t[j], sj_, ej_, esj);
                const int lo = max(si_, sj_), hi = min(ei_, ej_);
                if (hi > lo) {
                    const R* di = S.D + ((int64_t)L.k[i] * W + (lo - si_ + esi)) * F;
                    const R* dj = S.D + ((int64_t)L.k[j] * W + (lo - sj_ + esj)) * F;
                    for (int x = lane; x < (hi - lo) * F; x += 64) acc += (double)di[x] * (double)dj[x];
                    acc = wave_sum_f64(acc);
                }
                if (lane == 0) { L.g[i * kLocompMax + j] = acc; if (i == j) L.diag[i] = acc; }
            }
        }
        sy.full();
        // Cholesky of the Gram matrix by one wave (lane = row), right-looking; a pivot that vanishes against its own diagonal
        // marks an atom that the others already span: it keeps coefficient 0 (the pseudo-inverse would spread it)
        if (wv == 0) {
            for (int j = 0; j < n; ++j) {
                const double piv = L.g[j * kLocompMax + j];
                const bool dead = !(piv > kLocompDead * L.diag[j]);            // uniform
                const double ljj = dead ? 0.0 : sqrt(piv);
                if (lane == 0) L.g[j * kLocompMax + j] = ljj;
                if (lane > j && lane < n) {
                    const double lij = dead ? 0.0 : L.g[lane * kLocompMax + j] / ljj;
                    L.g[lane * kLocompMax + j] = lij;
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane > j && lane < n) {
                    const double lij = L.g[lane * kLocompMax + j];
                    for (int q = j + 1; q <= lane; ++q) L.g[lane * kLocompMax + q] -= lij * L.g[q * kLocompMax + j];
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (lane == 0) {
                // forward (L y = b), then backward (L^T x = y); a dead atom keeps 0
                for (int i = 0; i < n; ++i) {
                    double v = L.b[i];
                    for (int q = 0; q < i; ++q) v -= L.g[i * kLocompMax + q] * L.b[q];
                    const double d = L.g[i * kLocompMax + i];
                    L.b[i] = d > 0.0 ? v / d : 0.0;
                }
                for (int i = n - 1; i >= 0; --i) {
                    double v = L.b[i];
                    for (int q = i + 1; q < n; ++q) v -= L.g[q * kLocompMax + i] * L.b[q];
                    const double d = L.g[i * kLocompMax + i];
                    L.b[i] = d > 0.0 ? v / d : 0.0;
                }
                for (int i = 0; i < n; ++i) L.a[i] = (R)L.b[i];
            }
        }
        sy.full();
    }

    // ---- :1336-1341 / :1345-1350 coefficients += fitted; residual -= fitted * atom, with the local energies (:996-1016)
    for (int gi = 0; gi < n; ++gi) {
        const int tp = L.t[gi], kk = L.k[gi];
        const R cf = L.a[gi];
        if (tid == 0) {
            int si = L.si[gi];
            double before = 0.0;
            if (si < 0) { si = sh.nslots++; G.slot_t[si] = tp; G.slot_k[si] = kk; }
            else before = G.slot_a[si];
            const double after = before + (double)cf;
            G.slot_a[si] = after;
            if (gi == 0) {
                if (before != 0.0) sh.ndup += 1;
                const int ev = sh.nev++;
                G.ev_t[ev] = tp; G.ev_k[ev] = kk; G.ev_c[ev] = c;
            }
            sh.nnz += (after != 0.0 ? 1 : 0) - (before != 0.0 ? 1 : 0);
        }
        int s, e, es;
        const int len = centered_span(T, W, tp, s, e, es);
        R pb = (R)0, pa = (R)0;
        // (a policy that knows which cells can be non-zero updates only those -- and its lists of them)
        if (!Pol::update_residual(P, S, G, A, plds, tp, kk, cf, s, e, es, pb, pa)) {
            const int cnt = len * F;
            const R nc = -cf;
            const R* dk = S.D + ((int64_t)kk * W + es) * F;
            R* rv = G.r + (int64_t)s * F;
            constexpr int kU = 8;
            for (int i0 = tid; i0 < cnt; i0 += kThreads * kU) {
                R v[kU], d[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int i = i0 + u * kThreads;
                    v[u] = (R)0; d[u] = (R)0;
                    if (i < cnt) { v[u] = rv[i]; d[u] = dk[i]; }
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int i = i0 + u * kThreads;
                    if (i < cnt) {
                        const R sq = v[u] * v[u];
                        pb = pb + sq;
                        const R prod = nc * d[u];            // -c*D[k] rounded, then += (utils.py:120,129)
                        const R vn = v[u] + prod;
                        rv[i] = vn;
                        const R sq2 = vn * vn;
                        pa = pa + sq2;
                    }
                }
            }
        }
        pinned_tree2(pb, pa, sh.red, sy);
        if (tid == 0) { const R l = pb - pa; L.loss = L.loss + l; }     // :1005 summed over the group
        sy.full();                                                       // the residual writes of this atom are visible to the next
    }
    if (tid == 0) sh.e_res = sh.e_res - L.loss;                          // :1014
    if (P.has_scale)
        for (int gi = 0; gi < n; ++gi) {
            int s, e, es;
            centered_span(T, W, L.t[gi], s, e, es);
            const int sg0 = s >> P.seg_shift, sg1 = (e - 1) >> P.seg_shift;
            for (int sg = sg0 + wv; sg <= sg1; sg += kWaves) rscan_segment(P, G, sh, sg, lane);
        }

    // ---- :1353 re-correlation around every atom of the group (from the final residual), then the maxima of their segments
    Pol::before_runs(A, plds);
    sy.full();
    for (int gi = 0; gi < n; ++gi) {
        Pol::run(P, S, G, sh, A, plds, L.t[gi], L.k[gi]);
        sy.full();
    }
    for (int gi = 0; gi < n; ++gi) {
        const int tp = L.t[gi];
        const int lo = max(0, tp - (W - 1)), hi = min(T - 1, tp + (W - 1));
        const int sg0 = lo >> P.seg_shift, sg1 = hi >> P.seg_shift;
        if (P.blocked) {
            if (tid == 0) for (int sg = sg0; sg <= sg1; ++sg) sh.touched[sg >> 5] |= 1u << (sg & 31);
        } else {
            for (int sg = sg0 + wv; sg <= sg1; sg += kWaves) scan_segment<false>(P, G, wts, sh, sg, lane);
        }
    }

    // ---- :1357-1383 fast stop rules
    if (tid == 0) {
        sh.iters += 1;
        if ((double)sh.e_res < P.eps) { sh.converged = 1; sh.stop = STOP_ENERGY_EPS; }
        else if (P.l0 >= 0 && sh.nnz >= P.l0) { sh.converged = 1; sh.stop = STOP_NNZ; }
        else {
            bool done = false;
            if (P.has_snr) {
                const R q = sh.e_sig / sh.e_res;
                if ((double)q >= P.snr_ratio) { sh.converged = 1; sh.stop = STOP_SNR; done = true; }
            }
            if (!done) {
                const R delta = L.last_e - sh.e_res;
                if (fabs((double)delta) < P.eps) { sh.converged = 1; sh.stop = STOP_STALLED; }
            }
        }
    }
    sy.full();
}

}  // namespace hscmp
