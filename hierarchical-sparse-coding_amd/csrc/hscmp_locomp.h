// LoCOMP inside the greedy loop: the joint re-fit of a selected atom and its neighbourhood (hsc/modeling.py:1191-1425;
// Mailhe et al., ICASSP 2009) as the atom body of iterate_kernel.
//
// The reference's loop differs from ConvolutionalMatchingPursuit's (:1053-1186) in what happens to a selected atom
// (p, k, c):
//   :1222-1241  the previously selected atoms whose position lies within [start - W/2, end + W/2 (- 1)] of the atom's span
//               -- minus those that share its dictionary index, minus those whose row INSIDE THAT WINDOW equals the atom's
//               absolute position (the reference compares the two as they are; reproduced) -- form its group;
//   :1322-1341  the coefficients of the group are re-fitted on the local residual: least squares over the union of the
//               supports (np.linalg.pinv there; here the normal equations of the same system in float64 -- in registers for a
//               usual group, locomp_fast_solve; a diagonally pivoted, rank-revealing factorisation with the minimum-norm
//               completion, which is pinv's answer, for a rank-deficient or ill-conditioned one), ADDED to the stored
//               coefficients, and every atom of the group is removed from the residual and re-correlated around (:1343-1353);
//   :1368-1383  `coefficients.nnz` (stored non-zeros: an entry that cancels to 0.0 leaves the count) is what nbNonzeroCoefs
//               is compared with, and the loop also stops when an atom changes the residual energy by less than eps.
// Selection, weak-atom filter, residual subtraction with its local energies, local re-correlation and the round-level stop
// rules are those of the greedy loop and run through the same code.  Three policies: LocompRecorr (the table-free dense form, any
// shape), LocompSparse (multi-feature inputs with a sparse dictionary: hierarchical levels >= 1), LocompMfma (single-feature float32:
// re-correlations and the initial correlation on the matrix cores, up to four signals per workgroup around one dictionary image).
// The selections of a blocked round that lie far enough apart get all of that up to the fitted coefficients computed side by side,
// one wave each (locomp_precompute), before they are applied in order.
// Parity is at tolerance level by construction (the reference's pseudo-inverse is an SVD in the dictionary's dtype); the supports are
// exact on the reference's goldens (tests/test_locomp_hier.py: per-fixture tolerances from the conditioning its pseudo-inverse saw).
#pragma once
#include "hscmp_kernels.h"
#include "hscmp_mfma.h"
#include "hscmp_sparse.h"
#include "hscmp_rp_sparse.h"

namespace hscmp {

constexpr int kLocompMax = 128;       // atoms of a group whose LISTS stay in LDS, the selected one included (larger groups: the signal's global
                                      // scratch, up to DevParams::lg_cap atoms; beyond that STOP_GROUP and the host loop takes over)
constexpr int kLocompLds = 64;        // ... whose Gram matrix stays in LDS (larger groups: the signal's global scratch, Sig::lgram)
// Dependent atoms in a group (a Cholesky pivot that vanishes against its own diagonal entry): the group is rank deficient and the
// re-fit is the MINIMUM-NORM least-squares solution, which is what the reference's np.linalg.pinv (:1326) returns once its SVD has
// cut the vanishing singular value (locomp_atom: minimum-norm completion).  The re-fit runs in float64 whatever the dictionary's
// dtype.  The reference's cut-off is 1e-15 of the largest singular value IN THE DICTIONARY'S DTYPE: an exact dependency (a composite
// atom beside all of its singletons, a group that fills its signal) comes out of LAPACK as an exact 0 or ~1e-17 and is cut in both
// dtypes; a float32 group whose smallest singular value is mere round-off (1e-8 .. 1e-7 of the largest) is NOT cut there and the
// reference's coefficients are that round-off amplified by 1e7 (measured: DESIGN.md section 7e) -- nothing reproduces those.
constexpr double kLocompRankTol = 1e-14;      // x the largest diagonal entry: n u for the 64 .. 90 atoms of a large group (Higham's stopping rule);
                                              // exact dependencies leave 0 .. 2e-16 there, the nearly singular group of fuzz draw 3613 (sigma_min / sigma_max = 6e-8) 6e-14

template <typename R, int NMAX = kLocompMax, int NG = kLocompLds> struct LocompLds {
    int n, cnt;                       // group size; neighbours found (may exceed the capacity)
    int status, jout;                 // 2: the lists and fitted coefficients were computed ahead (locomp_precompute); jout: see jmin
    int jmin, jmax;                   // first / last position of the group; jout: an atom whose rows reach beyond the signal (their rows are re-correlated together otherwise)
    int pad_[2];
    int t[NMAX], k[NMAX], si[NMAX];       // position, atom, coefficient slot (-1: none yet), group order
    int ut[NMAX], uk[NMAX], usi[NMAX];    // neighbours as found (any order)
    R a[NMAX];                  // fitted coefficients in the dictionary's dtype (:1329)
    R loss, last_e;                   // energyLoss of the group (:998-1014), lastEnergyResidual (:1316)
    double b[NMAX];             // right-hand side <d_i, r>, then the solution
    double diag[NMAX];          // original diagonal (rank test)
    double g[NG * (NG + 1) / 2];          // Gram matrix <d_i, d_j> of the clipped atoms, then its Cholesky factor: lower
                                                       // triangle, row i at i (i + 1) / 2
                                          // (a group of more than NG atoms keeps it in global memory: Sig::lgram)
    static constexpr int kLdsGroup = NG;
    static constexpr int kFastGroup = NG >= 64 ? 32 : 16;   // atoms of a group that is re-fitted in registers (locomp_fast_solve)
    static __device__ __forceinline__ int at(int i, int j) { return i * (i + 1) / 2 + j; }       // (j <= i)
};

// the dense table-free loop (GenericRecorr) with the group re-fit as its atom body
template <typename R> struct LocompRecorr : GenericRecorr<R> {
    static constexpr bool kLocomp = true;
    static constexpr bool kWaveApply = true;            // (no per-policy residual update: short atoms go through one wave)
    static constexpr bool kGroupUpdate = false;
    static constexpr int kFastGroup = LocompLds<R>::kFastGroup;
    static constexpr bool kUnionRows = false;
    static constexpr bool kLoneRows = false;
    static constexpr bool kOwnInit = false;
    static constexpr int kMaxGroup = kLocompMax;
    using Lds = LocompLds<R>;
    using Base = GenericRecorr<R>;
    using Args = typename Base::Args;
    template <typename SH, typename SY>
    static __device__ __forceinline__ void lrun(const DevParams& P, const State<R>& S, const Sig<R>& G, SH& sh, const Args& A, char* lds, int p, int k, SY&)
    { Base::run(P, S, G, sh, A, lds, p, k); }
    template <typename SY>
    static __device__ __forceinline__ void lrun_span(const DevParams&, const State<R>&, const Sig<R>&, const Args&, char*, int, int, SY&) {}
    static __device__ __forceinline__ bool atom_lists(const DevParams&, const Args&, char*, const int*&, const int*&, const R*&) { return false; }
    static size_t extra_lds_bytes(const DevParams& P) { return Base::extra_lds_bytes(P) + sizeof(LocompLds<R>) + 16; }
    static __device__ __forceinline__ LocompLds<R>& group(const DevParams&, const Args&, char* lds)
    {
        return *reinterpret_cast<LocompLds<R>*>(lds + ((Base::kWinBytes + 15) / 16) * 16);
    }
    static __device__ __forceinline__ void before_runs(const Args&, char*) {}
    static __device__ __forceinline__ bool can_defer_rows(const DevParams&, const Args&, char*) { return false; }
    static __device__ __forceinline__ void rows_of_wave(const DevParams&, const State<R>&, const Sig<R>&, const Args&, char*, int, int) {}
};

// the same on the sparse policy (multi-feature inputs, sparse dictionary: hierarchical levels >= 1): the residual update keeps
// the per-row lists of non-zero cells current, the re-correlation forms the non-zero products only
template <typename R> struct LocompSparse : SparseRecorr<R, false> {
    static constexpr bool kLocomp = true;
    static constexpr int kMinWavesPerSimd = 2;          // (two workgroups per CU: the register-resident re-fit must not cost the second one)
    static constexpr int kFastGroup = 63;               // (the third level of BASELINE config 5 re-fits 30-60 atoms in one selection out of six:
                                                        //  this policy has the registers for the whole LDS-resident range; lane n holds b)
    static constexpr bool kWaveApply = false;           // (update_residual keeps the row lists: the workgroup form)
    static constexpr bool kUnionRows = true;
    static constexpr bool kLoneRows = true;             // (a lone interior atom too: computed ahead, committed by its wave, its rows through the per-wave pipeline --
                                                        //  three selections in four at BASELINE config 4's second level)
    static constexpr int kMaxGroup = kLocompMax;
    using Lds = LocompLds<R>;
    using Base = SparseRecorr<R, false>;
    using Args = typename Base::Args;
    template <typename SH, typename SY>
    static __device__ __forceinline__ void lrun(const DevParams& P, const State<R>& S, const Sig<R>& G, SH& sh, const Args& A, char* lds, int p, int k, SY&)
    { Base::run(P, S, G, sh, A, lds, p, k); }
    // the non-zeros of the atoms (k -> [nzptr[k], nzptr[k+1]) of (w << 16 | f, value)): right-hand sides and Gram entries of a group
    // are sums over a handful of them instead of W x F products
    static __device__ __forceinline__ bool atom_lists(const DevParams& P, const Args& A0, char* lds, const int*& nzptr, const int*& nzwf, const R*& nzval)
    {
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        nzptr = A.nzptr; nzwf = A.nzwf; nzval = A.nzval;
        return A.nzptr != nullptr;
    }
    // the four per-wave slots of the round-parallel loop's row pipeline, carved out of the policy's own LDS lists (nothing else uses them
    // between two atoms): available with row lists and by-feature dictionary lists at hand
    static __device__ __forceinline__ bool wave_slots(const Args& A0, const Args& A, size_t& sb, RpSparseCaps& c)
    {
        if (!(A0.rl_cnt && A.rl_cap == 8 && A.fptr && A.nzptr)) return false;
        sb = (sparse_lds_bytes<R>(A0.caps) / kWaves) & ~(size_t)15;
        c.nz = A0.caps.nz >= 512 ? 128 : 64;
        if (sb <= 16 + (size_t)c.nz * (sizeof(R) + 4)) return false;
        c.rec = (int)((sb - 16 - (size_t)c.nz * (sizeof(R) + 4)) / (8 + 3 * sizeof(R) + 8)) & ~7;
        return c.rec >= 64;
    }
    // rows [rb, re) of the union that starts at position pbase - (W - 1), by the calling wave in its own slot
    static __device__ __forceinline__ void rows_by_wave(const DevParams& P, const Sig<R>& G, const Args& A, char* lds, size_t sb, const RpSparseCaps& c,
                                                        int pbase, int rb, int re)
    {
        const int tid = ltid(), lane = tid & 63, wv = tid >> 6;
        const RpSparseSlot<R> SL = rp_sparse_slot<R>(lds + (size_t)wv * sb, c);
        int step = min(max(re - rb, 1), c.rec);
        for (int r0 = rb; r0 < re;) {
            const int nr = min(step, re - r0);
            if (RpSparse<R>::rows_listed(P, G, A, SL, c, pbase, r0, nr, true, 0, 0, lane)) { r0 += nr; continue; }
            if (nr > 1) { step = (nr + 1) / 2; continue; }
            RpSparse<R>::row_by_atoms(P, G, A, pbase, r0, true, 0, 0, lane);
            r0 += 1;
        }
    }
    // the calling wave's slot, lent to locomp_precompute between two uses by the row pipeline (nullptr: there are no slots)
    static __device__ __forceinline__ char* wave_scratch(const DevParams& P, const Args& A0, char* lds, size_t& bytes)
    {
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        size_t sb; RpSparseCaps c{};
        if (!wave_slots(A0, A, sb, c)) { bytes = 0; return nullptr; }
        bytes = sb;
        return lds + (size_t)(ltid() >> 6) * sb;
    }
    // deferred rows (locomp_rows_deferred): ALL rows of one selection's group by the wave that owns it
    static __device__ __forceinline__ bool can_defer_rows(const DevParams& P, const Args& A0, char* lds)
    {
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        size_t sb; RpSparseCaps c{};
        return (P.lc_ahead & 2) && wave_slots(A0, A, sb, c);
    }
    static __device__ __forceinline__ void rows_of_wave(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A0, char* lds, int pmin, int pmax)
    {
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        size_t sb; RpSparseCaps c{};
        wave_slots(A0, A, sb, c);
        rows_by_wave(P, G, A, lds, sb, c, pmin, 0, (pmax - pmin) + 2 * P.W - 1);
    }
    // the rows of a group of interior atoms between pmin and pmax, once: 2W-1 rows at a time (the capacity of sparse_rows), no
    // padding involved (every window lies inside the signal)
    template <typename SY>
    static __device__ __forceinline__ void lrun_span(const DevParams& P, const State<R>& S, const Sig<R>& G, const Args& A0, char* lds, int pmin, int pmax, SY&)
    {
        const SparseLds<R> L = sparse_lds_view<R>(lds, A0.caps);
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        // Row lists and by-feature dictionary lists at hand: the union of the group's rows in FOUR parts, one wave each, through the
        // per-wave window gather / pairing / sort / chains of the round-parallel level loop (RpSparse::rows_listed: the same pinned
        // chains, the same per-row arg-max) -- the workgroup-wide sparse_rows below walks the rows 2W - 1 at a time with a dozen
        // barriers per pass.  The four slots are carved out of the policy's own LDS lists (nothing else uses them between two atoms).
        size_t sb; RpSparseCaps c{};
        if ((P.lc_ahead & 2) && wave_slots(A0, A, sb, c)) {              // uniform
            const int wv = ltid() >> 6;
            const int nrows = (pmax - pmin) + 2 * P.W - 1;               // (row 0 = position pmin - (W - 1))
            const int per = (nrows + kWaves - 1) / kWaves;
            rows_by_wave(P, G, A, lds, sb, c, pmin, wv * per, min(nrows, wv * per + per));
            return;                                                      // (the caller's barrier follows)
        }
        unsigned* bits = Base::has_bits(P, A) ? Base::bits_of(P, A0, lds) : nullptr;
        if (bits) {                                          // (no row lists: the rows the group's subtractions may have filled)
            const int lo = max(0, pmin - (P.W - 1) / 2), hi = min(P.T, pmax + P.W / 2 + 1);
            for (int t = lo + (int)threadIdx.x; t < hi; t += kThreads) atomicOr(&bits[t >> 5], 1u << (t & 31));
        }                                                    // (ordered by the first barrier of gather_window)
        const int first = pmin - (P.W - 1), nrows = (pmax - pmin) + 2 * P.W - 1, step = 2 * P.W - 1;
        for (int r0 = 0; r0 < nrows; r0 += step)
            sparse_rows<R, false>(P, S, G, A, L, bits, first + r0, min(step, nrows - r0), false, 0, 0, false);
    }
    // The whole group applied in ONE pass (:1343-1350 removes its atoms one after the other).  The atoms of a level >= 1 dictionary
    // have a handful of non-zero cells each; the (cell, atom) pairs of the group are laid out in group order, a cell that several
    // atoms share is walked by the thread of its first pair in that order -- r = ((r + (-a0 d0)) + (-a1 d1)) ..., each product rounded
    // first: the reference's own sequence of roundings for that cell, and no cell depends on another.  The energy loss of the group
    // (:998-1014 sums before - after over the atoms' spans) is taken over the cells that change: unchanged cells cancel in it, and
    // what two atoms do to a shared cell telescopes -- the same number up to the rounding of the two span sums it no longer forms.
    // The cells are entered in the row lists as SparseRecorr::update_residual does.  Returns false (nothing done) without row lists /
    // dictionary lists or when the pairs do not fit the policy's LDS lists: the atom-by-atom form then runs.
    static constexpr bool kGroupUpdate = true;
    template <typename TF, typename KF, typename AF, typename SY>
    static __device__ __forceinline__ bool group_update(const DevParams& P, const Sig<R>& G, const Args& A0, char* lds, int n, TF T_, KF K_, AF A_,
                                                        R& loss, R* red, SY& sy)
    {
        if (!A0.rl_cnt) return false;
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        if (!A.nzptr || A.rl_cap != 8) return false;
        const SparseLds<R> L = sparse_lds_view<R>(lds, A0.caps);
        const int T = P.T, F = P.F, tid = ltid(), lane = tid & 63, wv = tid >> 6;
        if (n + 1 > L.caps.rec) return false;
        int* off = reinterpret_cast<int*>(L.okey);            // [n + 1] first pair of every atom
        int* key = L.perm;                                    // [pairs] cell (row * F + feature), -1 - q: clipped
        R* prod = L.rx;                                       // [pairs] -a * d, rounded (utils.py:120,129)
        // pairs per atom, prefix sum (one wave, 64 atoms at a time)
        if (wv == 0) {
            int base = 0;
            for (int g0 = 0; g0 < n; g0 += 64) {
                const int gi = g0 + lane;
                int c = 0;
                if (gi < n) { const int kk = K_(gi); c = A.nzptr[kk + 1] - A.nzptr[kk]; }
                int incl = c;
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m); if (lane >= m) incl += o; }
                if (gi < n) off[gi] = base + incl - c;
                base += __shfl(incl, 63);
            }
            if (lane == 0) { off[n] = base; L.ctl[0] = base; }
        }
        sy.lds();
        const int np = L.ctl[0];
        if (np > L.caps.rec) return false;                    // uniform
        for (int q = tid; q < np; q += kThreads) {
            int lo = 0, hi = n;                               // the atom of pair q: off[gi] <= q < off[gi + 1]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= q) lo = mid; else hi = mid; }
            const int gi = lo, kk = K_(gi), e = A.nzptr[kk] + (q - off[gi]);
            const int wf = A.nzwf[e], f = wf & 0xffff, g = T_(gi) - P.off + (wf >> 16);
            const R nc = -A_(gi);
            key[q] = (g >= 0 && g < T) ? g * F + f : -1 - q;  // clipped part of the atom (utils.py:110-129)
            prod[q] = nc * A.nzval[e];
        }
        sy.lds();
        int* cntw = A.rl_cnt + (int64_t)blockIdx.x * T;
        int* lfw = A.rl_f + (int64_t)blockIdx.x * T * 8;
        R mine = (R)0;
        for (int q = tid; q < np; q += kThreads) {
            const int cell = key[q];
            if (cell < 0) continue;
            bool first = true;
            for (int j = 0; j < q; ++j) first = first && key[j] != cell;
            if (!first) continue;
            const int g = cell / F, f = cell - g * F;
            // (both trips at once: the cell, its row's list)
            const R vb = G.r[cell];
            const int n2 = list_count(cntw + g);
            const int4* row2 = reinterpret_cast<const int4*>(lfw + (int64_t)g * 8);
            const int4 a2 = row2[0], b2 = row2[1];
            R v = vb + prod[q];
            for (int j = q + 1; j < np; ++j) if (key[j] == cell) v = v + prod[j];
            G.r[cell] = v;
            const R sb = vb * vb, sa = v * v;
            const R d = sb - sa;
            mine = mine + d;
            const bool listed = n2 > 8 || a2.x == f || a2.y == f || a2.z == f || a2.w == f || b2.x == f || b2.y == f || b2.z == f || b2.w == f;
            if (!listed) {
                const int o = atomicAdd(&cntw[g], 1);
                if (o < 8) lfw[(int64_t)g * 8 + o] = f;
            }
        }
        // the loss: lanes, then waves, in a fixed order
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mine = mine + __shfl_xor(mine, m);
        if (lane == 0) red[wv] = mine;
        sy.lds();
        loss = (red[0] + red[1]) + (red[2] + red[3]);
        sy.lds();
        return true;
    }
    // The same pass for ONE selection by ONE wave, ahead of its application (locomp_precompute): nothing is written -- lane l leaves with
    // the final values of the cells whose first pair is pair l or 64 + l, and the group's energy loss, summed exactly as group_update sums
    // it (threads 0..63 and 64..127 of the workgroup as two trees, then (p0 + p1) + (0 + 0)).  At most 128 pairs; scratch: the wave's own
    // LDS (off [n + 1], key / prod [128]).  Returns false when the lists are missing or too long: the application then runs group_update.
    static __device__ __forceinline__ bool cells_ahead(const DevParams& P, const Sig<R>& G, const Args& A0, char* lds, int n, const int* gt, const int* gk,
                                                       const double* ga, R c0, int* off, int* key, R* prod, int lane, int (&cell)[2], R (&val)[4], R& loss)
    {
        if (!A0.rl_cnt) return false;
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        if (!A.nzptr || A.rl_cap != 8) return false;
        const int T = P.T, F = P.F;
        auto wsync = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        int cnt = 0;
        if (lane < n) { const int kk = gk[lane]; cnt = A.nzptr[kk + 1] - A.nzptr[kk]; }
        int incl = cnt;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m); if (lane >= m) incl += o; }
        const int np = __shfl(incl, 63);
        if (np > 128) return false;                           // uniform
        if (lane < n) off[lane] = incl - cnt;
        if (lane == 0) off[n] = np;
        wsync();
        for (int q = lane; q < np; q += 64) {
            int lo = 0, hi = n;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= q) lo = mid; else hi = mid; }
            const int gi = lo, kk = gk[gi], e = A.nzptr[kk] + (q - off[gi]);
            const int wf = A.nzwf[e], f = wf & 0xffff, g = gt[gi] - P.off + (wf >> 16);
            const R nc = -(n > 1 ? (R)ga[gi] : c0);
            key[q] = (g >= 0 && g < T) ? g * F + f : -1 - q;
            prod[q] = nc * A.nzval[e];
        }
        wsync();
        R part[2] = {(R)0, (R)0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = lane + 64 * u;
            cell[u] = -1; val[u] = (R)0;
            if (q < np) {
                const int cl = key[q];
                bool first = cl >= 0;
                for (int j = 0; j < q && first; ++j) first = key[j] != cl;
                if (first) {
                    const R vb = G.r[cl];
                    R v = vb + prod[q];
                    for (int j = q + 1; j < np; ++j) if (key[j] == cl) v = v + prod[j];
                    const R sb = vb * vb, sa = v * v;
                    part[u] = sb - sa;
                    cell[u] = cl; val[u] = v;
                }
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { part[0] = part[0] + __shfl_xor(part[0], m); part[1] = part[1] + __shfl_xor(part[1], m); }
        const R z = (R)0;
        loss = (part[0] + part[1]) + (z + z);
        return true;
    }
    // ... and what is written when the selection is applied: the cells, and their entries in the row lists (as group_update does)
    static __device__ __forceinline__ void commit_cells(const DevParams& P, const Sig<R>& G, const Args& A0, char* lds, const int (&cell)[2], const R (&val)[4])
    {
        const Args A = dict_view(P, A0, lds + sparse_lds_bytes<R>(A0.caps));
        const int T = P.T, F = P.F;
        int* cntw = A.rl_cnt + (int64_t)blockIdx.x * T;
        int* lfw = A.rl_f + (int64_t)blockIdx.x * T * 8;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cl = cell[u];
            if (cl < 0) continue;
            const int g = cl / F, f = cl - g * F;
            const int n2 = list_count(cntw + g);
            const int4* row2 = reinterpret_cast<const int4*>(lfw + (int64_t)g * 8);
            const int4 a2 = row2[0], b2 = row2[1];
            G.r[cl] = val[u];
            const bool listed = n2 > 8 || a2.x == f || a2.y == f || a2.z == f || a2.w == f || b2.x == f || b2.y == f || b2.z == f || b2.w == f;
            if (!listed) {
                const int o = atomicAdd(&cntw[g], 1);
                if (o < 8) lfw[(int64_t)g * 8 + o] = f;
            }
        }
    }
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

// One 32-position tile against all atom groups: per-position best (coefficient, atom) straight from the accumulators -- the MFMA
// chain of an output IS the pinned sequential chain (hscmp_mfma.h: resolve_chain reproduces it bit for bit), so nothing has to be
// recomputed.  Per accumulator element: score, compare, three selects; ascending atoms within a lane and a strict '>' keep the lowest
// atom among equals, the two half-waves (same position, interleaved atom sets) merge with the same rule.  Result in every lane
// (position = lane & 31).  The greedy loop's tiles keep the score only (one v_max3 per two elements) because THEIR cost is the
// kernel's; here a tile is followed by nothing else.
template <int S4C, bool HAS_W>
__device__ __forceinline__ void mfma_tile_best(const float* __restrict__ dimg, const float* __restrict__ win, const float* __restrict__ wts,
                                               int G, int K, int lane, float& c_out, int& k_out)
{
    const int j = lane & 31, h = lane >> 5;
    const float* wb = win + j + h;
    constexpr int NM = 4 * S4C;
    float bop[NM];
#pragma unroll
    for (int s = 0; s < NM; ++s) bop[s] = wb[2 * s];
    const f32x4* dv = reinterpret_cast<const f32x4*>(dimg) + lane;
    float bs = -1.0f, bc = 0.0f;
    int bk = 0;
    for (int g = 0; g < G; ++g) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s4 = 0; s4 < S4C; ++s4) {
            const f32x4 a = dv[(g * S4C + s4) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bop[4 * s4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bop[4 * s4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], bop[4 * s4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], bop[4 * s4 + 3], acc, 0, 0, 0);
        }
        const int kbase = 32 * g + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kbase + (r & 3) + 8 * (r >> 2);                 // atom of accumulator element r (ascending in r)
            const float v = acc[r];
            float sc;
            if (HAS_W) { const float sw = v * wts[k]; sc = fabsf(sw); } else sc = fabsf(v);
            const bool take = k < K && sc > bs;
            bs = take ? sc : bs; bc = take ? v : bc; bk = take ? k : bk;
        }
    }
    const float os = swap_halves_f(bs, h), oc = swap_halves_f(bc, h);
    const int ok = swap_halves_i(bk, h);
    const bool take = os > bs || (os == bs && ok < bk);
    c_out = take ? oc : bc;
    k_out = take ? ok : bk;
}

// Single-feature float32 signals: the re-correlation of a group's rows on the matrix cores.  The loop keeps (coefficient, atom) per
// position like the dense form; mfma_tile_best hands both over for 32 rows at a time, bit-identical to the dense form's chains.
// GS signals per workgroup (1, 2 or 4) share ONE dictionary image: a CU then holds GS signals, and its matrix pipe has the other
// signals' tiles to run while one is in its serial steps (the 64 KB image leaves no room for a second workgroup); each signal's four
// waves meet at their own LDS counter (SoftSync), as in the four-signal greedy loop.
// LDS: [dictionary image | weights] then per signal [control block][window][LocompLds]
template <int S4C, bool HAS_W, int GS = 1> struct LocompMfma {
    using R = float;
    // four signals per workgroup: the per-signal state must fit a quarter of what the image leaves -- 256 segment maxima, the Gram
    // matrix of a group in LDS up to 32 atoms
    static constexpr int kMaxSegments = GS == 1 ? kMaxSeg : GS == 2 ? kMfmaMaxSeg : 256;
    static constexpr int kMaxGroup = kLocompMax;
    using Lds = LocompLds<float, kLocompMax, (GS <= 2 ? kLocompLds : 32)>;       // (four per workgroup: Gram matrices beyond 32 atoms in global memory)
    static constexpr bool kFused = false;
    static constexpr bool kLocomp = true;
    static constexpr bool kWaveApply = true;
    static constexpr bool kGroupUpdate = false;
    static constexpr int kFastGroup = Lds::kFastGroup;
    static constexpr int kMinWavesPerSimd = GS;
    static constexpr int kEnergyWaves = kWaves;
    static constexpr bool kScoreOnly = false;
    static constexpr int kGroup = GS;
    static constexpr int TP = 32;
    using Shared = IterSharedT<float, kMaxSegments, false, true>;        // (no merge buffers of the dense re-correlation)
    using Args = MfmaArgs;
    using Sync = typename std::conditional<GS == 1, HwSync, SoftSync>::type;
    static __device__ __forceinline__ Sync make_sync(Shared& sh)
    {
        if constexpr (GS == 1) return HwSync();
        else { SoftSync sy; sy.init(&sh.bar, &sh.bar_cnt); return sy; }
    }
    static __device__ __forceinline__ void epilogue(const DevParams&, const State<R>&, const Args&, char*, int) {}
    static __device__ __forceinline__ void resolve_wave(const DevParams&, const State<R>&, const Sig<R>&, const Args&, char*, int, int, int&, R&) {}
    static __device__ __forceinline__ const R* weights(const DevParams&, const State<R>& S, const Args&, char*) { return S.weights; }
    static __device__ __forceinline__ void on_atom(const DevParams&, const State<R>&, const Args&, char*, int, int) {}
    static __device__ __forceinline__ bool update_residual(const DevParams&, const State<R>&, const Sig<R>&, const Args&, char*, int, int, R,
                                                           int, int, int, R&, R&) { return false; }
    static __device__ __forceinline__ bool window_partials(const DevParams&, const Sig<R>&, const Args&, char*, int, int, R&) { return false; }
    static __device__ __forceinline__ bool wave_window_listed(const DevParams&, const Sig<R>&, const Args&, char*, int, int, int, int, R&) { return false; }
    static __device__ __forceinline__ bool row_results(const DevParams&, const Args&, char*, int, const int*&, const R*&, const R*&, int&, int&) { return false; }
    static __device__ __forceinline__ bool residual_copy_in_lds(const Args&, char*) { return false; }
    static __device__ __forceinline__ void before_runs(const Args&, char*) {}
    static __device__ __forceinline__ bool atom_lists(const DevParams&, const Args&, char*, const int*&, const int*&, const R*&) { return false; }

    // rows of one re-correlation: 2W-1 around one atom, or the union over a group of interior atoms -- their positions lie within
    // [p - W, p + W] of the selected atom (:1228-1236), so at most 4W-1 rows
    static constexpr bool kUnionRows = true;
    static constexpr bool kLoneRows = true;             // (an interior atom on its own: lrun_span's rows are lrun's)
    static __host__ __device__ int max_tiles(int W) { return (4 * W - 1 + TP - 1) / TP; }
    static __host__ __device__ int win_floats(int W) { return max_tiles(W) * TP + 8 * S4C + 32; }        // (a multiple of 4)
    static __host__ __device__ size_t front_bytes(int W) { return ((size_t)win_floats(W) * sizeof(float) + 15) / 16 * 16; }
    static __host__ __device__ size_t image_bytes(const Args& A) { return ((size_t)A.G * S4C * 256 + 32 * (size_t)A.G) * sizeof(float); }     // (x 16)
    static __host__ __device__ size_t policy_bytes(int W) { return front_bytes(W) + (sizeof(Lds) + 15) / 16 * 16; }
    static __host__ __device__ size_t per_signal_bytes(int W) { return ((sizeof(Shared) + 15) / 16) * 16 + policy_bytes(W); }
    static size_t total_lds_bytes(const DevParams& P, const Args& A) { return image_bytes(A) + (size_t)GS * per_signal_bytes(P.W); }
    static __device__ __forceinline__ int signal_lds_offset(const DevParams& P, const Args& A)
    {
        return __builtin_amdgcn_readfirstlane((int)image_bytes(A) + (GS == 1 ? 0 : gsig() * (int)per_signal_bytes(P.W)));
    }
    struct Lay { float* win; float* dimg; float* wts; int wf; };
    static __device__ __forceinline__ Lay layout(const DevParams& P, const Args& A, char* lds)
    {
        Lay L;
        L.wf = win_floats(P.W);
        L.win = reinterpret_cast<float*>(lds);
        L.dimg = reinterpret_cast<float*>(dyn_lds());
        L.wts = L.dimg + (size_t)A.G * S4C * 256;
        return L;
    }
    static __device__ __forceinline__ Lds& group(const DevParams& P, const Args&, char* lds)
    {
        return *reinterpret_cast<Lds*>(lds + front_bytes(P.W));
    }
    // the dictionary image and the weights, once per workgroup (all its threads, hardware barrier)
    static __device__ __forceinline__ void prologue_shared(const DevParams& P, const State<R>& S, const Args& A, char* smem)
    {
        float* dimg = reinterpret_cast<float*>(smem);
        float* wts = dimg + (size_t)A.G * S4C * 256;
        lds_copy16(dimg, A.dimg, A.G * S4C * 256 * (int)sizeof(float), (int)threadIdx.x, GS * kThreads);
        for (int i = threadIdx.x; i < 32 * A.G; i += GS * kThreads) wts[i] = (HAS_W && i < P.K) ? S.weights[i] : 0.0f;
        if (GS > 1 && ltid() == 0) {                          // the counter the signal's four waves meet at (SoftSync)
            Shared* sh = reinterpret_cast<Shared*>(smem + signal_lds_offset(P, A));
            sh->bar = 0u; sh->bar_cnt = 0;
        }
        __syncthreads();
    }
    static constexpr bool kOwnInit = true;      // the loop kernel computes the initial correlation itself (no launch in front of it)
    static __device__ __forceinline__ void prologue(const DevParams& P, const State<R>& S, const Args& A, char* lds, int b, Sync& sy)
    {
        // a fresh batch (no round yet): the initial correlation (:1293 convolve1d 'same': ZERO padded), 4W-1 rows at a time
        if (S.stats[(int64_t)b * ST_COUNT + ST_ROUNDS] == 0 && S.stats[(int64_t)b * ST_COUNT + ST_EVENTS] == 0) {
            Sig<R> G{};
            G.r = S.residual + (int64_t)b * P.T; G.bc = S.best_c + (int64_t)b * P.T; G.bk = S.best_k + (int64_t)b * P.T;
            const int step = 4 * P.W - 1;
            for (int t0 = 0; t0 < P.T; t0 += step) {
                rows(P, G, A, lds, t0, min(step, P.T - t0), false, 0, -1, sy);
                sy.full();
            }
        }
    }
    // rows [t0, t0 + nrows) from the residual window that starts at sample t0 - off: (coefficient, atom) of every row inside the
    // signal.  Samples outside it: none when `interior`; else reflected about the slice [sidx, sidx + nslice) (np.pad 'reflect',
    // :1046), or zero (nslice < 0: the initial correlation, :159-164)
    static __device__ __forceinline__ void rows(const DevParams& P, const Sig<R>& G, const Args& A, char* lds, int t0, int nrows,
                                                bool interior, int sidx, int nslice, Sync& sy, bool timed = false)
    {
        const Lay L = layout(P, A, lds);
        const int T = P.T, W = P.W, tid = ltid(), lane = tid & 63, wv = tid >> 6;
        const int span = nrows + W - 1, wstart = t0 - P.off;
        HSCMP_STAMP_BEGIN();
        for (int i = tid; i < L.wf; i += kThreads) {
            float v = 0.0f;
            if (i < span) {
                const int gi = wstart + i;
                if (interior) v = G.r[gi];
                else if (nslice < 0) v = (gi >= 0 && gi < T) ? G.r[gi] : 0.0f;
                else v = G.r[reflect_index(gi, sidx, nslice)];
            }
            L.win[i] = v;
        }
        sy.lds();
        if (timed) HSCMP_STAMP(10);                            // window
        const int nt = (nrows + TP - 1) / TP;
        for (int q = wv; q < nt; q += kWaves) {
            float c; int k;
            mfma_tile_best<S4C, HAS_W>(L.dimg, L.win + TP * q, L.wts, A.G, P.K, lane, c, k);
            const int row = TP * q + lane, t = t0 + row;
            if (lane < TP && row < nrows && t >= 0 && t < T) { G.bc[t] = c; G.bk[t] = k; }      // overlapReplace clipping (utils.py:133-161)
        }
        if (timed) HSCMP_STAMP(11);                            // tiles: (coefficient, atom) of the rows
    }
    // rows p-(W-1) .. p+(W-1) around one atom (:1018-1051)
    template <typename SH>
    static __device__ __forceinline__ void lrun(const DevParams& P, const State<R>&, const Sig<R>& G, SH&, const Args& A, char* lds, int p, int, Sync& sy)
    {
        const int T = P.T, W = P.W;
        const int tstart = p - P.off - (W - 1), tend = p + W / 2 + (W - 1);          // :1028-1038
        const int sidx = tstart < 0 ? 0 : tstart, eidx = tend > T - 1 ? T - 1 : tend;
        rows(P, G, A, lds, p - (W - 1), 2 * W - 1, tstart >= 0 && tend <= T - 1, sidx, eidx - sidx + 1, sy, true);
    }
    // the rows of a group of atoms between pmin and pmax, none of which reaches a signal end: a row's value only depends on the
    // final residual, so the union of their row ranges is computed once
    static __device__ __forceinline__ void lrun_span(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A, char* lds, int pmin, int pmax, Sync& sy)
    {
        rows(P, G, A, lds, pmin - (P.W - 1), (pmax - pmin) + 2 * P.W - 1, true, 0, 0, sy, true);
    }
    // deferred rows (locomp_rows_deferred): the same rows by ONE wave -- its window in the wave's quarter of the group state (idle between
    // two batches), its tiles one after the other
    static constexpr size_t kWaveStride = (sizeof(Lds) / kWaves) & ~(size_t)15;
    static __device__ __forceinline__ bool can_defer_rows(const DevParams& P, const Args&, char*) { return (size_t)win_floats(P.W) * sizeof(float) <= kWaveStride; }
    static __device__ __forceinline__ void rows_of_wave(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A, char* lds, int pmin, int pmax)
    {
        const Lay L = layout(P, A, lds);
        const int W = P.W, tid = ltid(), lane = tid & 63, wv = tid >> 6;
        const int t0 = pmin - (W - 1), nrows = (pmax - pmin) + 2 * W - 1;
        const int span = nrows + W - 1, wstart = t0 - P.off;
        float* win = reinterpret_cast<float*>(reinterpret_cast<char*>(&group(P, A, lds)) + (size_t)wv * kWaveStride);
        for (int i = lane; i < L.wf; i += 64) win[i] = i < span ? G.r[wstart + i] : 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nt = (nrows + TP - 1) / TP;
        for (int q = 0; q < nt; ++q) {
            float c; int k;
            mfma_tile_best<S4C, HAS_W>(L.dimg, win + TP * q, L.wts, A.G, P.K, lane, c, k);
            const int row = TP * q + lane;
            if (lane < TP && row < nrows) { G.bc[t0 + row] = c; G.bk[t0 + row] = k; }       // (interior atoms: every row lies inside the signal)
        }
    }
    // (the step-by-step atom body of the greedy loop is never instantiated for a kLocomp policy, but must compile)
    template <typename SH>
    static __device__ __forceinline__ void run(const DevParams&, const State<R>&, const Sig<R>&, SH&, const Args&, char*, int, int) {}
};

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ---- a group beyond its LDS copy: the signal's scratch in global memory (Sig::lgram, lgram_doubles(cap) doubles per signal) ----
// [Gram matrix / its factor: lower triangle, cap (cap + 1) / 2][S = L^T L of the minimum-norm completion: the same][b][diag][w][a]
// [t][k][si][ut][uk][usi] (ints).  All accesses are agent-scope atomics (L2), as the slot hash table's: whatever a wave's vector
// cache holds, a value written before a barrier is the value read behind it.
struct GroupGlobal {
    double* gram; double* s; double* b; double* diag; double* w; double* a;
    int* t; int* k; int* si; int* ut; int* uk; int* usi;
};
__device__ __forceinline__ GroupGlobal group_global(double* base, int cap)
{
    GroupGlobal g;
    const int64_t tri = (int64_t)cap * (cap + 1) / 2;
    g.gram = base; g.s = base + tri; g.b = g.s + tri; g.diag = g.b + cap; g.w = g.diag + cap; g.a = g.w + cap;
    int* ib = reinterpret_cast<int*>(g.a + cap);
    g.t = ib; g.k = ib + cap; g.si = ib + 2 * cap; g.ut = ib + 3 * cap; g.uk = ib + 4 * cap; g.usi = ib + 5 * cap;
    return g;
}
template <typename T> __device__ __forceinline__ T gld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void gst(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// arg-max of (value, index) over the lanes of a wave, the lowest index among equal values; the result in every lane
__device__ __forceinline__ void wave_argmax_f64(double& v, int& i)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(v, m);
        const int oi = __shfl_xor(i, m);
        const bool take = ov > v || (ov == v && oi < i);
        v = take ? ov : v; i = take ? oi : i;
    }
}
__device__ __forceinline__ void wave_argmax_f64(double& v, int& i, int& aux)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(v, m);
        const int oi = __shfl_xor(i, m), oa = __shfl_xor(aux, m);
        const bool take = ov > v || (ov == v && oi < i);
        v = take ? ov : v; i = take ? oi : i; aux = take ? oa : aux;
    }
}

// (row, column <= row) of entry e of a lower triangle stored row by row
__device__ __forceinline__ void tri_decode(int e, int& i, int& j)
{
    i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    while (i * (i + 1) / 2 > e) --i;
    j = e - i * (i + 1) / 2;
}

// Cholesky of a symmetric positive semi-definite matrix by ALL threads of the signal (the wave form below is the faster one for
// the handful of atoms of a usual group; this one for a factor that lives in global memory): right-looking, per column the scaled
// column, a barrier, the trailing triangle spread over the threads, a barrier.  dead(j, pivot) marks a column the earlier ones
// already span: it is skipped -- its row keeps the coefficients that express it through them, its own column is zero -- which IS
// the factor G = L L^T of the semi-definite matrix with the zero columns left in place.  Returns the number of dead columns.
template <typename GET, typename SET, typename DEAD, typename SY>
__device__ __forceinline__ int wg_cholesky(int n, GET get, SET set, DEAD dead_test, SY& sy)
{
    const int tid = ltid();
    int nd = 0;
    for (int j = 0; j < n; ++j) {
        const double piv = get(j, j);
        const bool dead = dead_test(j, piv);                       // uniform
        const double ljj = dead ? 0.0 : sqrt(piv);
        for (int r = j + 1 + tid; r < n; r += kThreads) set(r, j, dead ? 0.0 : get(r, j) / ljj);
        sy.full();
        if (tid == 0) set(j, j, ljj);
        if (!dead) {
            const int M = n - j - 1, tot = M * (M + 1) / 2;
            for (int e = tid; e < tot; e += kThreads) {
                int ri, ci;
                tri_decode(e, ri, ci);
                const int r = j + 1 + ri, q = j + 1 + ci;
                set(r, q, get(r, q) - get(r, j) * get(q, j));
            }
        }
        sy.full();
        nd += dead ? 1 : 0;
    }
    return nd;
}
// x = (L L^T)^-1 v in place, L a lower factor with no (or explicitly zero) dead diagonal entries; column-oriented, one barrier per
// column: every thread reads v[i] and L[i][i], the rows behind (in front of) i take their update, thread 0 stores the result
template <typename GET, typename VGET, typename VSET, typename SY>
__device__ __forceinline__ void wg_solve(int n, GET get, VGET vget, VSET vset, SY& sy)
{
    const int tid = ltid();
    for (int i = 0; i < n; ++i) {                                  // L y = v
        const double d = get(i, i), yi = d > 0.0 ? vget(i) / d : 0.0;
        for (int r = i + 1 + tid; r < n; r += kThreads) vset(r, vget(r) - get(r, i) * yi);
        sy.full();
        if (tid == 0) vset(i, yi);
    }
    sy.full();
    for (int i = n - 1; i >= 0; --i) {                             // L^T x = y
        const double d = get(i, i), xi = d > 0.0 ? vget(i) / d : 0.0;
        for (int q = tid; q < i; q += kThreads) vset(q, vget(q) - get(i, q) * xi);
        sy.full();
        if (tid == 0) vset(i, xi);
    }
    sy.full();
}

// ---- the re-fit of a usual group (a handful of independent atoms) entirely in registers ------------------------------------------
// One wave; lane r < n holds row r of the Gram matrix (G[r][0..r]), lane n the right-hand side as one more row: the column
// operations of the Cholesky factorisation then leave y = L^-1 b in that lane (the factor of the matrix bordered by b), and the
// backward substitution walks the rows of L through v_readlane.  Register indices are static (the loops are unrolled to NF with
// uniform guards), lane indices uniform: no LDS traffic, no wave barrier, no dependent memory round trip -- the LDS form took 84 k
// (10 atoms) to 237 k (18 atoms) cycles per selection at BASELINE config 5, most of it the single-lane substitutions.
// No pivoting: a pivot below kLocompFastPivot of its atom's own norm (an ill-conditioned or rank-deficient group) returns false and
// the group goes through the pivoted, rank-revealing path.  g: packed lower triangle (left as it is), b: right-hand sides.
constexpr double kLocompFastPivot = 1e-6;
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(reinterpret_cast<const char*>(p) - dyn_lds()); }
template <int NF>
__device__ __attribute__((noinline)) bool locomp_fast_solve(unsigned g_off, unsigned b_off, int n_, unsigned x_off)
{
    // (byte offsets into the workgroup's dynamic LDS, not pointers: a generic pointer to LDS across this call boundary costs flat
    //  accesses and, with the offsets of locomp_precompute's workspaces, trips the backend -- "Illegal instruction ... src_shared_base")
    const double* g = reinterpret_cast<const double*>(dyn_lds() + g_off);
    const double* b = reinterpret_cast<const double*>(dyn_lds() + b_off);
    double* x_out = reinterpret_cast<double*>(dyn_lds() + x_off);
    const int lane = (int)(threadIdx.x & 63u);
    const int n = __builtin_amdgcn_readfirstlane(n_);
    double row[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        double v = 0.0;
        if (j < n) {
            if (lane < n && j <= lane) v = g[lane * (lane + 1) / 2 + j];
            else if (lane == n) v = b[j];
        }
        row[j] = v;
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        if (j < n && ok) {                                        // uniform
            const double piv = readlane_f64(row[j], j);
            if (!(piv > kLocompFastPivot * g[j * (j + 1) / 2 + j])) ok = false;      // (against the atom's own norm: the original diagonal entry)
            else {
                const double ljj = sqrt(piv);
                const double lrj = lane == j ? ljj : row[j] / ljj;
                if (lane >= j) row[j] = lrj;
                // (NF > 32: no guard per column -- thousands of tiny basic blocks are what the compiler chokes on; a column beyond n
                //  is all zeros in every lane that would take the update, except entry n of the b row, which nobody reads)
#pragma unroll
                for (int q = j + 1; q < NF; ++q)
                    if (NF > 32 || q < n) {
                        const double lqj = readlane_f64(row[j], q);
                        if (lane >= q) row[q] -= lrj * lqj;
                    }
            }
        }
    }
    if (!ok) return false;
    // L^T x = y: y sits in lane n; x_r = y_r / L[r][r], then y_i -= L[r][i] x_r for the rows in front of r
#pragma unroll
    for (int r = NF - 1; r >= 0; --r) {
        if (r < n) {
            const double lrr = readlane_f64(row[r], r);
            const double yr = readlane_f64(row[r], n);
            const double xr = yr / lrr;
            if (lane == n) row[r] = xr;
#pragma unroll
            for (int i = 0; i < r; ++i) {
                const double lri = readlane_f64(row[i], r);
                if (lane == n) row[i] -= lri * xr;
            }
        }
    }
    if (lane == n) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
            if (i < n) x_out[i] = row[i];
    }
    return true;
}

// sum over x = sub, sub + 8, ... < cnt of a[x] * b[x] in float64, x ascending (one lane of the eight that share an item of the normal
// equations): the loads of four steps are issued together -- one memory round trip instead of four -- the sums stay in order
template <typename R>
__device__ __forceinline__ double strided8_dot(const R* __restrict__ a, const R* __restrict__ b, int sub, int cnt)
{
    double acc = 0.0;
    for (int x0 = sub; x0 < cnt; x0 += 32) {
        R av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int x = x0 + 8 * u; av[u] = (R)0; bv[u] = (R)0; if (x < cnt) { av[u] = a[x]; bv[u] = b[x]; } }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (x0 + 8 * u < cnt) acc += (double)av[u] * (double)bv[u];
    }
    return acc;
}

// ---- the selections of a blocked round side by side, as far as they do not depend on each other -------------------------------
// The atoms of a blocked round (:908-963) are applied one after the other (:1314), but when they lie more than 5W + 8 samples apart
// nothing one of them does reaches what another one reads: neighbours come from positions within W of the selected atom, their spans
// reach 1.5 W, the rows re-correlated for a group lie within 2 W and are formed from samples within 2.5 W -- against residual and
// coefficient slots changed within 1.5 W.  (BASELINE configs 4 / 5: ten blocks over 65536 samples, 6 500 samples apart.)  Three things
// follow, each switched by a bit of HSCMP_LOCOMP_AHEAD and each bit-identical to the selection-by-selection loop:
//  * locomp_precompute: neighbourhood, normal equations, re-fit -- and the subtractions with their local energies, on a private copy of
//    the span (dense) or as a list of final cell values (sparse) -- of up to four selections AT ONCE, one wave each, before any of them is
//    applied.  Nothing is written but wave-private LDS (four workspaces that alias the group state Pol::Lds; the sparse policy lends each
//    wave its slot of the row pipeline for the Gram matrix: groups of up to 54 atoms); results leave through registers.
//  * locomp_atom's first branch: such a group is COMMITTED BY ITS WAVE ALONE when its turn comes -- residual, coefficient slots, event,
//    counters, segment marks, stop rules -- and the workgroup meets once per selection; a round that stops half-way drops the rest.
//  * locomp_rows_deferred: the rows of the batch are re-correlated behind its last selection (or as the first job of the next batch's
//    computations ahead), one wave per selection.
// (GRAM_INSIDE false: the Gram matrix lives in scratch the policy lends the wave -- Pol::wave_scratch -- and the lists may be longer)
template <typename R, int NW, bool GRAM_INSIDE = true> struct WaveGroup {
    int n, pad_;
    int t[NW], k[NW], si[NW], ut[NW], uk[NW], usi[NW];
    double b[NW];
    double g[GRAM_INSIDE ? NW * (NW + 1) / 2 : 1];
    R a[NW];
};
template <typename R, typename Pol, typename SH, typename SY>
__device__ __forceinline__ void locomp_precompute(const DevParams& P, const State<R>& S, const Sig<R>& G, SH& sh, const typename Pol::Args& A, char* plds,
                                                  const int* ord_t, const int* ord_k, const R* ord_c, int first, int count, LocompPre<R>& pre,
                                                  LocompRows& rows, SY& sy)
{
    // sparse policy: the Gram matrix (and the pair lists after it) in the wave's slot of the row pipeline, 12 KB at BASELINE config 5 -- groups
    // of up to 54 atoms then (one selection out of six of its third level has more than 32)
    constexpr bool kLent = Pol::kGroupUpdate;
    constexpr int NW = kLent ? 64 : Pol::kFastGroup >= 32 ? 32 : 16;
    using WG = WaveGroup<R, NW, !kLent>;
    constexpr size_t kStride = (sizeof(typename Pol::Lds) / kWaves) & ~(size_t)15;
    static_assert(sizeof(WG) <= kStride, "four wave workspaces must fit the group state they alias");
    typename Pol::Lds& L = Pol::group(P, A, plds);
    const int T = P.T, W = P.W, F = P.F, tid = ltid(), lane = tid & 63, wv = tid >> 6;
    pre.status = 0; pre.n = 0; pre.t = 0; pre.k = 0; pre.si = -1; pre.a = (R)0; pre.u0 = 0; pre.ulen = 0; pre.loss = (R)0;
    pre.span[0] = pre.span[1] = pre.span[2] = pre.span[3] = (R)0; pre.cell[0] = pre.cell[1] = -1;
    auto wave_sync = [&]() {                                     // LDS written by a lane of this wave, read by another one
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // rows of the previous batch that still wait (locomp_atom, may_defer): this wave's first (its window, where the policy needs one, lies in
    // the wave's own workspace)
    if constexpr (Pol::kUnionRows) if (rows.pending) { Pol::rows_of_wave(P, S, G, A, plds, rows.pmin, rows.pmax); rows.pending = 0; }
    if (wv < count) {
        WG& w = *reinterpret_cast<WG*>(reinterpret_cast<char*>(&L) + (size_t)wv * kStride);
        const int p = ord_t[first + wv], k = ord_k[first + wv];
        const R c = ord_c[first + wv];
        int s0, e0, es0;
        centered_span(T, W, p, s0, e0, es0);
        const int nstart = max(s0 - W / 2, 0);
        const int nend = min((e0 - 1) + ((W & 1) ? W / 2 : W / 2 - 1), T);
        const int nlast = min(nend, T - 1);
        if (lane == 0) { w.n = 0; w.t[0] = p; w.k[0] = k; w.si[0] = -1; }
        wave_sync();
        for (int ti = nstart + lane; ti <= nlast; ti += 64)          // :1222-1241 through the per-position slot chains
            for (int i = hval_load(G.head + ti); i >= 0; i = hval_load(G.hval + i)) {
                const int ki = G.slot_k[i];
                if (ti == p && ki == k) w.si[0] = i;
                if (ki == k || (ti - nstart) == p) continue;
                if (!(G.slot_a[i] != 0.0)) continue;
                const int o = atomicAdd(&w.n, 1);
                if (o < NW - 1) { w.ut[o] = ti; w.uk[o] = ki; w.usi[o] = i; }
            }
        wave_sync();
        const int m = __builtin_amdgcn_readfirstlane(w.n);
        double* wg = w.g;                                            // the Gram matrix of this wave's group
        size_t wg_bytes = sizeof(w.g);
        if constexpr (kLent) {
            wg = reinterpret_cast<double*>(Pol::wave_scratch(P, A, plds, wg_bytes));
            if (!wg) {                                               // (no slots: what the workspace leaves behind the lists -- 27 atoms)
                constexpr size_t used = (sizeof(WG) + 15) & ~(size_t)15;
                wg = reinterpret_cast<double*>(reinterpret_cast<char*>(&w) + used);
                wg_bytes = kStride - used;
            }
        }
        // (a larger group: the selection goes the usual way)
        bool ok = m <= NW - 1 && wg != nullptr && (size_t)(m + 1) * (m + 2) / 2 * sizeof(double) <= wg_bytes;
        if (ok) {
            if (lane < m) {                                          // group order: (position, atom) ascending behind the selected atom
                const long long key = ((long long)w.ut[lane] << 32) | (unsigned)w.uk[lane];
                int rank = 0;
                for (int q = 0; q < m; ++q) rank += ((((long long)w.ut[q] << 32) | (unsigned)w.uk[q]) < key) ? 1 : 0;
                w.t[1 + rank] = w.ut[lane]; w.k[1 + rank] = w.uk[lane]; w.si[1 + rank] = w.usi[lane];
            }
            wave_sync();
            const int n = 1 + m;
            if (n > 1) {
                const int nitems = n + n * (n + 1) / 2;
                const int* nzp = nullptr; const int* nzwf = nullptr; const R* nzv = nullptr;
                if (Pol::atom_lists(P, A, plds, nzp, nzwf, nzv)) {       // uniform: sparse dictionary, one lane per item
                    for (int it = lane; it < nitems; it += 64) {
                        double acc = 0.0;
                        if (it < n) {
                            const int ti = w.t[it] - P.off, ki = w.k[it];
                            for (int e = nzp[ki]; e < nzp[ki + 1]; ++e) {
                                const int wf = nzwf[e], row = ti + (wf >> 16);
                                if (row >= 0 && row < T) acc += (double)nzv[e] * (double)G.r[(int64_t)row * F + (wf & 0xffff)];
                            }
                            w.b[it] = acc;
                        } else {
                            int i, j;
                            tri_decode(it - n, i, j);
                            const int ti = w.t[i] - P.off, tj = w.t[j] - P.off, ki = w.k[i], kj = w.k[j];
                            const int ej0 = nzp[kj], ej1 = nzp[kj + 1];
                            for (int e = nzp[ki]; e < nzp[ki + 1]; ++e) {
                                const int wf = nzwf[e], row = ti + (wf >> 16);
                                if (row < 0 || row >= T) continue;
                                const int want = ((row - tj) << 16) | (wf & 0xffff);
                                if (row - tj < 0 || row - tj >= W) continue;
                                for (int e2 = ej0; e2 < ej1; ++e2)
                                    if (nzwf[e2] == want) acc += (double)nzv[e] * (double)nzv[e2];
                            }
                            wg[i * (i + 1) / 2 + j] = acc;
                        }
                    }
                } else {                                                 // dense dictionary: eight lanes per item
                    for (int it = lane >> 3; it < nitems; it += 8) {
                        const int sub = lane & 7;
                        double acc = 0.0;
                        int i = 0, j = 0;
                        if (it < n) {
                            int s_, e_, es_;
                            const int len = centered_span(T, W, w.t[it], s_, e_, es_);
                            const R* dk = S.D + ((int64_t)w.k[it] * W + es_) * F;
                            const R* rv = G.r + (int64_t)s_ * F;
                            acc = strided8_dot(dk, rv, sub, len * F);
                        } else {
                            tri_decode(it - n, i, j);
                            int si_, ei_, esi, sj_, ej_, esj;
                            centered_span(T, W, w.t[i], si_, ei_, esi);
                            centered_span(T, W, w.t[j], sj_, ej_, esj);
                            const int lo = max(si_, sj_), hi = min(ei_, ej_);
                            if (hi > lo) {
                                const R* di = S.D + ((int64_t)w.k[i] * W + (lo - si_ + esi)) * F;
                                const R* dj = S.D + ((int64_t)w.k[j] * W + (lo - sj_ + esj)) * F;
                                acc = strided8_dot(di, dj, sub, (hi - lo) * F);
                            }
                        }
                        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
                        if (sub == 0) { if (it < n) w.b[it] = acc; else wg[i * (i + 1) / 2 + j] = acc; }
                    }
                }
                wave_sync();
                if (n <= 8) ok = locomp_fast_solve<8>(lds_off(wg), lds_off(w.b), n, lds_off(w.b));
                else if (n <= 16) ok = locomp_fast_solve<16>(lds_off(wg), lds_off(w.b), n, lds_off(w.b));
                else if (!kLent || n <= 32) ok = locomp_fast_solve<(kLent ? 32 : NW)>(lds_off(wg), lds_off(w.b), n, lds_off(w.b));
                else ok = locomp_fast_solve<Pol::kFastGroup>(lds_off(wg), lds_off(w.b), n, lds_off(w.b));
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                wave_sync();
            }
            if (ok) {
                pre.status = 2; pre.n = n;
                if (lane < n) { pre.t = w.t[lane]; pre.k = w.k[lane]; pre.si = w.si[lane]; pre.a = n > 1 ? (R)w.b[lane] : c; }
            }
            // Sparse dictionary with row lists: the group's cells and its energy loss too (LocompSparse::cells_ahead), in the LDS the normal
            // equations have left; applied by the owning wave when the selection's turn comes.
            if constexpr (Pol::kGroupUpdate) {
                constexpr size_t kOffBytes = ((NW + 1) * sizeof(int) + 15) & ~(size_t)15;            // off[n + 1], then prod[128], key[128]
                if (ok && wg_bytes >= kOffBytes + 128 * (sizeof(int) + sizeof(R))) {
                    int* off = reinterpret_cast<int*>(wg);
                    R* prod = reinterpret_cast<R*>(reinterpret_cast<char*>(wg) + kOffBytes);
                    int* key = reinterpret_cast<int*>(prod + 128);
                    R loss;
                    if (Pol::cells_ahead(P, G, A, plds, n, w.t, w.k, w.b, c, off, key, prod, lane, pre.cell, pre.span, loss)) {
                        pre.status = 3; pre.loss = loss;
                    }
                }
            }
            // Dense dictionary, short atoms: the group's subtractions too, on a PRIVATE copy of the stretch of the residual they touch
            // (in the LDS the normal equations have left; at most 256 samples, which then travel in four registers per lane) -- atom
            // after atom with the local energies of :996-1016, the same products, sums and trees as apply on the residual itself.
            if constexpr (Pol::kWaveApply) {
                if (ok && W * F <= 1024) {
                    int s_lo = T, e_hi = 0;
                    if (lane < n) { int s_, e_, es_; centered_span(T, W, w.t[lane], s_, e_, es_); s_lo = s_; e_hi = e_; }
#pragma unroll
                    for (int m_ = 32; m_ >= 1; m_ >>= 1) { s_lo = min(s_lo, __shfl_xor(s_lo, m_)); e_hi = max(e_hi, __shfl_xor(e_hi, m_)); }
                    const int u0 = s_lo, ulen = (e_hi - s_lo) * F;
                    if (ulen <= 256 && (size_t)ulen * sizeof(R) <= sizeof(w.g)) {
                        R* cp = reinterpret_cast<R*>(w.g);
                        const R* src = G.r + (int64_t)u0 * F;
                        for (int i = lane; i < ulen; i += 64) cp[i] = src[i];
                        wave_sync();
                        R loss = (R)0;
                        // (an atom's elements are fetched while its predecessor is applied: cnt <= ulen <= 256, four per lane)
                        R dn[4];
                        auto fetch_atom = [&](int gi) {
                            int s, e, es;
                            const int cnt = centered_span(T, W, w.t[gi], s, e, es) * F;
                            const R* dk = S.D + ((int64_t)w.k[gi] * W + es) * F;
#pragma unroll
                            for (int u = 0; u < 4; ++u) { const int i = lane + 64 * u; dn[u] = i < cnt ? dk[i] : (R)0; }
                        };
                        fetch_atom(0);
                        for (int gi = 0; gi < n; ++gi) {
                            const int tp = w.t[gi];
                            const R cf = n > 1 ? (R)w.b[gi] : c;
                            int s, e, es;
                            const int cnt = centered_span(T, W, tp, s, e, es) * F;
                            const R nc = -cf;
                            R* rv = cp + (s - u0) * F;
                            R b4[4] = {(R)0, (R)0, (R)0, (R)0}, a4[4] = {(R)0, (R)0, (R)0, (R)0};
                            R v[4], d[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) { const int i = lane + 64 * u; d[u] = dn[u]; v[u] = i < cnt ? rv[i] : (R)0; }
                            if (gi + 1 < n) fetch_atom(gi + 1);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int i = lane + 64 * u;
                                if (i < cnt) {
                                    const R sq = v[u] * v[u];
                                    b4[u] = b4[u] + sq;
                                    const R prod = nc * d[u];            // -c*D[k] rounded, then += (utils.py:120,129)
                                    const R vn = v[u] + prod;
                                    rv[i] = vn;
                                    const R sq2 = vn * vn;
                                    a4[u] = a4[u] + sq2;
                                }
                            }
#pragma unroll
                            for (int m_ = 32; m_ >= 1; m_ >>= 1) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const R ob = __shfl_down(b4[u], m_), oa = __shfl_down(a4[u], m_);
                                    b4[u] = b4[u] + ob; a4[u] = a4[u] + oa;
                                }
                            }
                            {
                                const R b01 = b4[0] + b4[1], b23 = b4[2] + b4[3], a01 = a4[0] + a4[1], a23 = a4[2] + a4[3];
                                const R pb = b01 + b23, pa = a01 + a23;
                                const R l = pb - pa;
                                loss = loss + l;                         // (lane 0 holds the group's sum, :1005)
                            }
                            wave_sync();
                        }
                        pre.status = 3; pre.u0 = u0; pre.ulen = ulen; pre.loss = __shfl(loss, 0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int i = lane + 64 * u; pre.span[u] = i < ulen ? cp[i] : (R)0; }
                    }
                }
            }
        }
    }
    sy.full();
}

// One selected atom (p, k, c): modeling.py:1314-1383.  All threads of the signal's workgroup; the caller leaves the atom
// loop when sh.skip or sh.converged is set afterwards.
template <typename R, typename Pol, typename SH, typename SY>
__device__ __forceinline__ int locomp_atom(const DevParams& P, const State<R>& S, const Sig<R>& G, SH& sh, const typename Pol::Args& A,
                                            char* plds, const R* wts, int p, int k, R c, SY& sy, const LocompPre<R>& pre, int owner,
                                            bool may_defer, LocompRows& rows)
{
    const int T = P.T, W = P.W, F = P.F, tid = ltid(), lane = tid & 63, wv = tid >> 6;
    typename Pol::Lds& L = Pol::group(P, A, plds);
    constexpr int kCap = Pol::kMaxGroup;
    const GroupGlobal GG = group_global(G.lgram, P.lg_cap);
    HSCMP_STAMP_BEGIN();
    // lists and fitted coefficients computed ahead (locomp_precompute; owner: the wave whose registers hold them)?
    bool have = false, applied_ahead = false;
    if (owner >= 0) {                                            // uniform
        if (wv == owner) {
            int st = pre.status;
            // A group that was applied ahead and whose rows can wait is COMMITTED BY ITS WAVE ALONE -- residual, coefficient slots, event,
            // counters, segment marks, stop rules, everything the steps below do for it, from the wave's registers -- and the workgroup
            // meets once per selection instead of six times.
            if (st == 3 && may_defer && !P.has_scale) {
                const int n = pre.n;
                const bool mine = lane < n;
                const int tp = pre.t;
                const int pmn = wave_min_i32(mine ? tp : INT_MAX), pmx = wave_max_i32(mine ? tp : INT_MIN);
                const bool outside = mine && !(tp - P.off - (W - 1) >= 0 && tp + W / 2 + (W - 1) <= T - 1);      // (padding in its rows, :1028-1046)
                if ((n > 1 || Pol::kLoneRows) && __ballot(outside) == 0 && pmx - pmn <= 2 * W) {
                    bool full = false;
                    if (lane == 0 && sh.nev >= P.cap) { sh.converged = 1; sh.stop = STOP_CAPACITY; sh.skip = 1; full = true; }
                    full = __shfl(full ? 1 : 0, 0) != 0;
                    if (!full) {
                        const R last_e = sh.e_res;                                   // :1316
                        if constexpr (Pol::kGroupUpdate) Pol::commit_cells(P, G, A, plds, pre.cell, pre.span);
                        else {
                            R* dst = G.r + (int64_t)pre.u0 * F;
#pragma unroll
                            for (int u = 0; u < 4; ++u) { const int i = lane + 64 * u; if (i < pre.ulen) dst[i] = pre.span[u]; }
                        }
                        if (mine) {                                                  // :1336-1341, :1368 (the bookkeeping loop below, lane = atom)
                            const int kk = pre.k;
                            int si = pre.si;
                            double before = 0.0;
                            if (si < 0) {                                            // (lane 0 only)
                                si = sh.nslots++; G.slot_t[si] = tp; G.slot_k[si] = kk;
                                hval_store(G.hval + si, hval_load(G.head + tp)); hval_store(G.head + tp, si);
                            }
                            else before = G.slot_a[si];
                            const double after = before + (double)pre.a;
                            G.slot_a[si] = after;
                            if (lane == 0) {
                                if (before != 0.0) sh.ndup += 1;
                                const int ev = sh.nev++;
                                G.ev_t[ev] = tp; G.ev_k[ev] = kk; G.ev_c[ev] = c;
                            }
                            const int d = (after != 0.0 ? 1 : 0) - (before != 0.0 ? 1 : 0);
                            if (d) atomicAdd(&sh.nnz, d);
                            const int sg0 = max(0, tp - (W - 1)) >> P.seg_shift, sg1 = min(T - 1, tp + (W - 1)) >> P.seg_shift;
                            for (int sg = sg0; sg <= sg1; ++sg) atomicOr(&sh.touched[sg >> 5], 1u << (sg & 31));
                        }
                        if (lane == 0) {                                             // :1014, :1357-1383
                            sh.e_res = sh.e_res - pre.loss;
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
                                    const R delta = last_e - sh.e_res;
                                    if (fabs((double)delta) < P.eps) { sh.converged = 1; sh.stop = STOP_STALLED; }
                                }
                            }
                        }
                        rows.pending = 1; rows.pmin = pmn; rows.pmax = pmx;
                    }
                    if (lane == 0) sh.lc_flag[owner & 1] = 5 | ((full || sh.converged) ? 2 : 0);
                    st = 4;
                }
            }
            if (st != 4 && lane == 0) sh.lc_flag[owner & 1] = 0;
            if (st == 2 || st == 3) {
                if (lane < pre.n) { L.t[lane] = pre.t; L.k[lane] = pre.k; L.si[lane] = pre.si; L.a[lane] = pre.a; }
                bool full = false;
                if (lane == 0) {
                    L.n = pre.n; L.cnt = pre.n - 1; L.loss = st == 3 ? pre.loss : (R)0; L.last_e = sh.e_res;
                    L.jmin = INT_MAX; L.jmax = INT_MIN; L.jout = 0;
                    if (sh.nev >= P.cap) { sh.converged = 1; sh.stop = STOP_CAPACITY; sh.skip = 1; full = true; }
                }
                full = __shfl(full ? 1 : 0, 0) != 0;
                if (st == 3 && !full) {
                    if constexpr (Pol::kGroupUpdate) Pol::commit_cells(P, G, A, plds, pre.cell, pre.span);      // the group's cells and their row-list entries
                    else {                                       // the residual of the group's span, as its atoms left it on the private copy
                        R* dst = G.r + (int64_t)pre.u0 * F;
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int i = lane + 64 * u; if (i < pre.ulen) dst[i] = pre.span[u]; }
                    }
                }
            }
            if (lane == 0 && st != 4) L.status = st;
        }
        sy.full();
        // committed by its wave; its rows wait (locomp_rows_deferred).  Nothing else of the shared state is read here: the next selection's
        // wave may be writing it already
        if (const int fl = sh.lc_flag[owner & 1]) return fl;
        if (sh.skip) return 0;
        have = L.status >= 2;
        applied_ahead = L.status == 3;
    }
    // ---- event list, the atom's own entry, its neighbourhood (:1222-1241)
    if (!have) {
        if (tid == 0) {
            if (sh.nev >= P.cap) { sh.converged = 1; sh.stop = STOP_CAPACITY; sh.skip = 1; }
            L.cnt = 0; L.t[0] = p; L.k[0] = k; L.si[0] = -1; L.loss = (R)0; L.last_e = sh.e_res;
            L.jmin = INT_MAX; L.jmax = INT_MIN; L.jout = 0;
        }
        sy.full();
        if (sh.skip) return 0;
    }
    int s0, e0, es0;
    centered_span(T, W, p, s0, e0, es0);
    const int nstart = max(s0 - W / 2, 0);
    const int nend = min((e0 - 1) + ((W & 1) ? W / 2 : W / 2 - 1), T);
    // the coefficient slots of a position are chained (head[t]: the most recent one, hval[slot]: the one before it; iterate_kernel's
    // prologue builds the chains, the bookkeeping below extends them): the neighbourhood is a walk over the chains of its 2W + 1
    // positions, one thread each, instead of a scan of the whole slot list (14 k slots per signal at BASELINE config 5)
    const int nlast = min(nend, T - 1);
    if (!have) {
        for (int ti = nstart + tid; ti <= nlast; ti += kThreads)
            for (int i = hval_load(G.head + ti); i >= 0; i = hval_load(G.hval + i)) {
                const int ki = G.slot_k[i];
                if (ti == p && ki == k) L.si[0] = i;                 // (at most one)
                if (ki == k || (ti - nstart) == p) continue;
                if (!(G.slot_a[i] != 0.0)) continue;                  // (the list-of-lists matrix drops an entry that became 0.0)
                const int o = atomicAdd(&L.cnt, 1);
                if (o < kCap - 1) { L.ut[o] = ti; L.uk[o] = ki; L.usi[o] = i; }
            }
        sy.full();
    }
    HSCMP_STAMP(0);                                              // neighbourhood scan
    const int m = L.cnt;
    if (m > P.lg_cap - 1) {                                      // uniform: beyond the signal's scratch too (hscmp_params / HSCMP_LOCOMP_GROUP_CAP)
        if (tid == 0) { sh.converged = 1; sh.stop = STOP_GROUP; sh.skip = 1; }
        sy.full();
        return 0;
    }
    // more neighbours than the LDS lists hold: the lists of this group live in the signal's global scratch
    const bool bigL = m > kCap - 1;                              // uniform
    if (bigL) {
        sy.full();                                               // (everybody has read the count)
        if (tid == 0) L.cnt = 0;
        sy.full();
        for (int ti = nstart + tid; ti <= nlast; ti += kThreads)
            for (int i = hval_load(G.head + ti); i >= 0; i = hval_load(G.hval + i)) {
                const int ki = G.slot_k[i];
                if (ki == k || (ti - nstart) == p) continue;
                if (!(G.slot_a[i] != 0.0)) continue;
                const int o = atomicAdd(&L.cnt, 1);
                gst(GG.ut + o, ti); gst(GG.uk + o, ki); gst(GG.usi + o, i);
            }
        sy.full();
    }
    auto T_ = [&](int i) -> int { return bigL ? gld(GG.t + i) : L.t[i]; };
    auto K_ = [&](int i) -> int { return bigL ? gld(GG.k + i) : L.k[i]; };
    auto SI_ = [&](int i) -> int { return bigL ? gld(GG.si + i) : L.si[i]; };
    auto A_ = [&](int i) -> R { return bigL ? (R)gld(GG.a + i) : L.a[i]; };
    auto setA = [&](int i, R v) { if (bigL) gst(GG.a + i, (double)v); else L.a[i] = v; };
    auto B_ = [&](int i) -> double { return bigL ? gld(GG.b + i) : L.b[i]; };
    auto setB = [&](int i, double v) { if (bigL) gst(GG.b + i, v); else L.b[i] = v; };
    auto Dg_ = [&](int i) -> double { return bigL ? gld(GG.diag + i) : L.diag[i]; };
    auto setDg = [&](int i, double v) { if (bigL) gst(GG.diag + i, v); else L.diag[i] = v; };
    auto UT_ = [&](int i) -> int { return bigL ? gld(GG.ut + i) : L.ut[i]; };
    auto setUT = [&](int i, int v) { if (bigL) gst(GG.ut + i, v); else L.ut[i] = v; };
    // group order: the new atom, then the neighbours by (position, atom) -- the order of the reference's sparse slice
    if (!have)
    for (int i = tid; i < m; i += kThreads) {
        const int ui = UT_(i), uki = bigL ? gld(GG.uk + i) : L.uk[i], usi = bigL ? gld(GG.usi + i) : L.usi[i];
        const long long key = ((long long)ui << 32) | (unsigned)uki;
        int rank = 0;
        for (int q = 0; q < m; ++q) {
            const int uq = UT_(q), ukq = bigL ? gld(GG.uk + q) : L.uk[q];
            rank += ((((long long)uq << 32) | (unsigned)ukq) < key) ? 1 : 0;
        }
        if (bigL) { gst(GG.t + 1 + rank, ui); gst(GG.k + 1 + rank, uki); gst(GG.si + 1 + rank, usi); }
        else { L.t[1 + rank] = ui; L.k[1 + rank] = uki; L.si[1 + rank] = usi; }
    }
    if (!have) {
        if (tid == 0) {
            L.n = 1 + m;
            if (bigL) { gst(GG.t, p); gst(GG.k, k); gst(GG.si, L.si[0]); gst(GG.a, (double)c); }
            else L.a[0] = c;
        }
        sy.full();
    }
    const int n = 1 + m;
    HSCMP_STAMP(1);                                              // group order
#ifdef HSCMP_DBG_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_stamps[14] += 1; g_stamps[15] += (unsigned long long)n; g_stamps[16 + min(n >> 3, 15)] += 1; }
#endif

    // the Gram matrix: in LDS, or -- a group larger than the policy keeps there -- in the signal's global scratch (through L2)
    const bool big = n > Pol::Lds::kLdsGroup;                // uniform
    double* gglob = GG.gram;
    auto gl = [&](int idx) -> double { return big ? gld(gglob + idx) : L.g[idx]; };
    auto gs = [&](int idx, double v) { if (big) gst(gglob + idx, v); else L.g[idx] = v; };
    if (n > 1 && !have) {
        // ---- :1322-1329 least squares of the local residual on the group's (clipped) atoms: G x = b, float64
        // items: n right-hand sides, then the n (n + 1) / 2 Gram entries; one wave per item, lanes over the elements
        const int nitems = n + n * (n + 1) / 2;
        const int* nzp = nullptr; const int* nzwf = nullptr; const R* nzv = nullptr;
        if (Pol::atom_lists(P, A, plds, nzp, nzwf, nzv)) {           // uniform
            // sparse dictionary: one thread per item, sums over the atoms' non-zeros (a clipped atom loses the rows outside the signal)
            for (int it = tid; it < nitems; it += kThreads) {
                double acc = 0.0;
                if (it < n) {
                    const int ti = T_(it) - P.off, ki = K_(it);
                    for (int e = nzp[ki]; e < nzp[ki + 1]; ++e) {
                        const int wf = nzwf[e], row = ti + (wf >> 16);
                        if (row >= 0 && row < T) acc += (double)nzv[e] * (double)G.r[(int64_t)row * F + (wf & 0xffff)];
                    }
                    setB(it, acc);
                } else {
                    int i, j;
                    tri_decode(it - n, i, j);
                    const int ti = T_(i) - P.off, tj = T_(j) - P.off, ki = K_(i), kj = K_(j);
                    const int ej0 = nzp[kj], ej1 = nzp[kj + 1];
                    for (int e = nzp[ki]; e < nzp[ki + 1]; ++e) {
                        const int wf = nzwf[e], row = ti + (wf >> 16);
                        if (row < 0 || row >= T) continue;
                        const int want = ((row - tj) << 16) | (wf & 0xffff);             // the same cell seen from atom j
                        if (row - tj < 0 || row - tj >= W) continue;
                        for (int e2 = ej0; e2 < ej1; ++e2)
                            if (nzwf[e2] == want) acc += (double)nzv[e] * (double)nzv[e2];
                    }
                    gs(L.at(i, j), acc);
                    if (i == j) setDg(i, acc);
                }
            }
        } else if (nitems <= 4 * kThreads) {
            // dense dictionary, a usual group: EIGHT LANES per item (32 items per pass), each a strided share of the overlap, three
            // exchange steps -- a wave per item left 56 lanes idle on a 9-item group and made 65 items sixteen passes; a thread per
            // item made every sum a chain of 64 dependent loads
            for (int it = tid >> 3; it < nitems; it += kThreads >> 3) {
                const int sub = tid & 7;
                double acc = 0.0;
                int i = 0, j = 0;
                if (it < n) {
                    int s_, e_, es_;
                    const int len = centered_span(T, W, T_(it), s_, e_, es_);
                    const R* dk = S.D + ((int64_t)K_(it) * W + es_) * F;
                    const R* rv = G.r + (int64_t)s_ * F;
                    acc = strided8_dot(dk, rv, sub, len * F);
                } else {
                    tri_decode(it - n, i, j);
                    int si_, ei_, esi, sj_, ej_, esj;
                    centered_span(T, W, T_(i), si_, ei_, esi);
                    centered_span(T, W, T_(j), sj_, ej_, esj);
                    const int lo = max(si_, sj_), hi = min(ei_, ej_);
                    if (hi > lo) {
                        const R* di = S.D + ((int64_t)K_(i) * W + (lo - si_ + esi)) * F;
                        const R* dj = S.D + ((int64_t)K_(j) * W + (lo - sj_ + esj)) * F;
                        acc = strided8_dot(di, dj, sub, (hi - lo) * F);
                    }
                }
                acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
                if (sub == 0) {
                    if (it < n) setB(it, acc);
                    else { gs(L.at(i, j), acc); if (i == j) setDg(i, acc); }
                }
            }
        } else
        for (int it = wv; it < nitems; it += kWaves) {
            double acc = 0.0;
            if (it < n) {
                int s, e, es;
                const int len = centered_span(T, W, T_(it), s, e, es);
                const R* dk = S.D + ((int64_t)K_(it) * W + es) * F;
                const R* rv = G.r + (int64_t)s * F;
                for (int i = lane; i < len * F; i += 64) acc += (double)dk[i] * (double)rv[i];
                acc = wave_sum_f64(acc);
                if (lane == 0) setB(it, acc);
            } else {
                // (row i, column j <= i) from the linear index of the lower triangle
                int i, j;
                tri_decode(it - n, i, j);
                int si_, ei_, esi, sj_, ej_, esj;
                centered_span(T, W, T_(i), si_, ei_, esi);
                centered_span(T, W, T_(j), sj_, ej_, esj);
                const int lo = max(si_, sj_), hi = min(ei_, ej_);
                if (hi > lo) {
                    const R* di = S.D + ((int64_t)K_(i) * W + (lo - si_ + esi)) * F;
                    const R* dj = S.D + ((int64_t)K_(j) * W + (lo - sj_ + esj)) * F;
                    for (int x = lane; x < (hi - lo) * F; x += 64) acc += (double)di[x] * (double)dj[x];
                    acc = wave_sum_f64(acc);
                }
                if (lane == 0) { gs(L.at(i, j), acc); if (i == j) setDg(i, acc); }
            }
        }
        sy.full();
        HSCMP_STAMP(2);                                          // right-hand sides + Gram entries
        // Cholesky WITH DIAGONAL PIVOTING of the Gram matrix: at every step the largest remaining diagonal entry of the Schur
        // complement is the pivot; the factorisation ends when that entry no longer exceeds kLocompRankTol x the largest original
        // one -- what is left then is round-off RELATIVE TO THE MATRIX (Higham, "Analysis of the Cholesky decomposition of a
        // semi-definite matrix"), whatever the conditioning of the atoms already taken: without pivoting the Schur complement of a
        // dependent atom carries eps x cond of its predecessors, 4e-12 of its own norm in one of the reference's goldens.  Pivoting
        // is virtual: the pivot order perm[] / pos[] is recorded, L[i][step s] stays at the packed entry (i, perm[s]) of the
        // symmetric matrix, which the Schur complement no longer needs.  All atoms taken (every group of independent atoms):
        // x = G^-1 b by two substitutions in pivot order.  Otherwise the group is rank deficient and the reference's np.linalg.pinv
        // (:1326), which cuts the vanishing singular values, returns the MINIMUM-NORM least-squares solution; so does the
        // completion below.
        auto sym = [&](int i, int j) -> int { return i >= j ? L.at(i, j) : L.at(j, i); };
        auto PERM_ = UT_;                                          // pivot of step s            (the free list `ut`)
        auto POS_ = [&](int i) -> int { return bigL ? gld(GG.usi + i) : L.usi[i]; };      // step at which atom i was taken, -1: not (yet)
        auto setPOS = [&](int i, int v) { if (bigL) gst(GG.usi + i, v); else L.usi[i] = v; };
        int rank = -1;
        constexpr int kFast = Pol::kFastGroup;
        if (!big && n <= kFast) {                                  // uniform: the usual group, in registers (locomp_fast_solve)
            if (wv == 0) {
                // (three unrollings: the guards of the unused steps are what a small group would pay for)
                bool ok;
                if (n <= 8) ok = locomp_fast_solve<8>(lds_off(L.g), lds_off(L.b), n, lds_off(L.b));
                else if (kFast >= 16 && n <= 16) ok = locomp_fast_solve<16>(lds_off(L.g), lds_off(L.b), n, lds_off(L.b));
                else if (kFast > 32 && n <= 32) ok = locomp_fast_solve<32>(lds_off(L.g), lds_off(L.b), n, lds_off(L.b));
                else ok = locomp_fast_solve<kFast>(lds_off(L.g), lds_off(L.b), n, lds_off(L.b));
                if (lane == 0) L.cnt = ok ? n : -1;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (ok && lane < n) L.a[lane] = (R)L.b[lane];
            }
            sy.full();
            rank = L.cnt;
        }
        if (rank >= 0) {
        } else if (!big) {
            // one wave, lane = atom (n <= 64: the Gram matrix is in LDS)
            if (wv == 0) {
                int rk = 0;
                unsigned long long done = 0ull;
                double dmax0 = 0.0;
                double* col = reinterpret_cast<double*>(L.uk);           // (the free list `uk` and what follows it: NMAX ints = 64 doubles)
                if (lane < n) { L.usi[lane] = -1; L.diag[lane] = L.b[lane]; }       // (diag: the right-hand side under elimination)
                for (int s_ = 0; s_ < n; ++s_) {
                    double v = (lane < n && !((done >> lane) & 1ull)) ? L.g[L.at(lane, lane)] : -1.0;
                    int idx = lane;
                    wave_argmax_f64(v, idx);
                    if (s_ == 0) dmax0 = v;
                    if (!(v > kLocompRankTol * dmax0)) break;                     // uniform
                    const int pp = idx;
                    const double ljj = sqrt(v);
                    const double ys = L.diag[pp] / ljj;                           // y of this step (L y = b, row by row as L appears)
                    if (lane == 0) { L.g[L.at(pp, pp)] = ljj; L.ut[s_] = pp; L.usi[pp] = s_; L.diag[pp] = ys; }
                    done |= 1ull << pp;
                    const bool mine = lane < n && !((done >> lane) & 1ull);
                    double lrp = 0.0;
                    if (mine) { lrp = L.g[sym(lane, pp)] / ljj; L.g[sym(lane, pp)] = lrp; L.diag[lane] -= lrp * ys; }
                    // the scaled column as a compact array (0 for the atoms already taken: their entries of a row hold L and must
                    // stay -- x - l * 0 leaves them as they are), so that the trailing update of a row is one branch-free sweep
                    if (lane < n) col[lane] = lrp;
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (mine) {
                        double* rowp = L.g + L.at(lane, 0);
                        for (int q = 0; q <= lane; ++q) rowp[q] -= lrp * col[q];
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    ++rk;
                }
                if (lane == 0) L.cnt = rk;
                if (rk == n) {
                    // backward (L^T z = y) in pivot order, the atoms taken before step s in parallel; x[perm[s]] = z[s]
                    const int mypos = lane < n ? L.usi[lane] : 0x7fffffff;
                    for (int s_ = n - 1; s_ >= 0; --s_) {
                        const int p2 = L.ut[s_];
                        const double z = L.diag[p2] / L.g[L.at(p2, p2)];
                        if (mypos < s_) L.diag[lane] -= L.g[sym(p2, lane)] * z;
                        if (lane == 0) { L.b[p2] = z; L.a[p2] = (R)z; }
                        __builtin_amdgcn_wave_barrier();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                }
            }
            sy.full();
            rank = L.cnt;
        } else {
            // all threads, the matrix in global memory; the atoms not yet taken as a list (`uk`), so that the trailing update
            // walks the pairs of what is left
            auto ACT_ = [&](int i) -> int { return bigL ? gld(GG.uk + i) : L.uk[i]; };
            auto setACT = [&](int i, int v) { if (bigL) gst(GG.uk + i, v); else L.uk[i] = v; };
            for (int i = tid; i < n; i += kThreads) { setPOS(i, -1); setACT(i, i); }
            sy.full();
            int rk = 0, na = n;
            double dmax0 = 0.0;
            for (int s_ = 0; s_ < n; ++s_) {
                double v = -1.0;
                int idx = 0x7fffffff, slot = -1;
                for (int a_ = lane; a_ < na; a_ += 64) {
                    const int i = ACT_(a_);
                    const double d = gl(L.at(i, i));
                    if (d > v || (d == v && i < idx)) { v = d; idx = i; slot = a_; }
                }
                wave_argmax_f64(v, idx, slot);                     // (every wave finds the same pivot)
                if (s_ == 0) dmax0 = v;
                if (!(v > kLocompRankTol * dmax0)) break;           // uniform
                const int pp = idx;
                const double ljj = sqrt(v);
                const int last = ACT_(na - 1);
                sy.full();                                         // (everybody has read the diagonal and the list)
                if (tid == 0) { gs(L.at(pp, pp), ljj); setUT(s_, pp); setPOS(pp, s_); setACT(slot, last); }
                --na;
                sy.full();
                for (int a_ = tid; a_ < na; a_ += kThreads) { const int r = ACT_(a_); gs(sym(r, pp), gl(sym(r, pp)) / ljj); }
                sy.full();
                for (int e = tid; e < na * (na + 1) / 2; e += kThreads) {
                    int ai, bi;
                    tri_decode(e, ai, bi);
                    const int r = ACT_(ai), q = ACT_(bi);
                    gs(sym(r, q), gl(sym(r, q)) - gl(sym(r, pp)) * gl(sym(q, pp)));
                }
                sy.full();
                ++rk;
            }
            rank = rk;
            if (rank == n) {
                for (int i = tid; i < n; i += kThreads) setDg(i, B_(PERM_(i)));
                sy.full();
                wg_solve(n, [&](int i, int j) { return gl(sym(PERM_(i), PERM_(j))); }, Dg_, setDg, sy);
                for (int i = tid; i < n; i += kThreads) { const int pi = PERM_(i); const double x = Dg_(i); setB(pi, x); setA(pi, (R)x); }
            }
        }
        if (rank < n) {
            // ---- minimum-norm solution x = G^+ b of the rank-deficient group.  G = L L^T with L [n x rank] of full column rank
            // (row i: the atom, column s: the pivot step; an atom taken at step s' has zeros behind column s'), so
            // G^+ = L (L^T L)^-2 L^T:  u = L^T b,  S = L^T L,  v = S^-1 u,  w = S^-1 v,  x = L w.
            // Rare (a composite atom together with all of its singletons, a group that fills its whole signal): all threads,
            // S in the signal's global scratch.
            sy.full();
            auto Lf = [&](int i, int s_) -> double { const int ps = POS_(i); return (ps < 0 || ps >= s_) ? gl(sym(i, PERM_(s_))) : 0.0; };
            auto sget = [&](int i, int j) -> double { return gld(GG.s + (int64_t)i * (i + 1) / 2 + j); };
            auto sset = [&](int i, int j, double v) { gst(GG.s + (int64_t)i * (i + 1) / 2 + j, v); };
            for (int it = tid; it < rank + rank * (rank + 1) / 2; it += kThreads) {
                double acc = 0.0;
                if (it < rank) {
                    for (int i = 0; i < n; ++i) acc += Lf(i, it) * B_(i);
                    gst(GG.w + it, acc);                                               // u
                } else {
                    int pi, qi;
                    tri_decode(it - rank, pi, qi);
                    for (int i = 0; i < n; ++i) acc += Lf(i, pi) * Lf(i, qi);
                    sset(pi, qi, acc);
                }
            }
            sy.full();
            wg_cholesky(rank, sget, sset, [&](int, double piv) { return !(piv > 0.0); }, sy);
            auto wget = [&](int i) -> double { return gld(GG.w + i); };
            auto wset = [&](int i, double v) { gst(GG.w + i, v); };
            wg_solve(rank, sget, wget, wset, sy);
            wg_solve(rank, sget, wget, wset, sy);
            for (int i = tid; i < n; i += kThreads) {
                double x = 0.0;
                for (int q = 0; q < rank; ++q) x += Lf(i, q) * wget(q);
                setB(i, x); setA(i, (R)x);
            }
        }
        sy.full();
        HSCMP_STAMP(3);                                          // Cholesky + substitutions
    }

    // ---- :1336-1341 / :1345-1350 coefficients += fitted; residual -= fitted * atom, with the local energies (:996-1016)
    // the coefficient slots of the group, all atoms side by side (only the selected atom can be new to the list; it alone leaves an
    // event); nnz counts stored non-zeros (:1368)
    for (int gi = tid; gi < n; gi += kThreads) {
        const int tp = T_(gi), kk = K_(gi);
        const R cf = A_(gi);
        int si = SI_(gi);
        double before = 0.0;
        if (si < 0) {                                                                   // (gi == 0 only)
            si = sh.nslots++; G.slot_t[si] = tp; G.slot_k[si] = kk;
            hval_store(G.hval + si, hval_load(G.head + tp)); hval_store(G.head + tp, si);       // the new head of its position's chain
        }
        else before = G.slot_a[si];
        const double after = before + (double)cf;
        G.slot_a[si] = after;
        if (gi == 0) {
            if (before != 0.0) sh.ndup += 1;
            const int ev = sh.nev++;
            G.ev_t[ev] = tp; G.ev_k[ev] = kk; G.ev_c[ev] = c;
        }
        const int d = (after != 0.0 ? 1 : 0) - (before != 0.0 ? 1 : 0);
        if (d) atomicAdd(&sh.nnz, d);
        if constexpr (Pol::kUnionRows) {                                                // (read behind the barriers below)
            atomicMin(&L.jmin, tp); atomicMax(&L.jmax, tp);
            if (!(tp - P.off - (W - 1) >= 0 && tp + W / 2 + (W - 1) <= T - 1)) L.jout = 1;     // interior: no padding in its rows (:1028-1046)
        }
    }
    // A policy that knows the cells of its atoms (sparse dictionary, row lists) applies the whole group in one pass
    bool grouped = false;
    if constexpr (Pol::kGroupUpdate) if (!applied_ahead) {
        R gloss = (R)0;
        grouped = Pol::group_update(P, G, A, plds, n, T_, K_, A_, gloss, sh.red, sy);        // uniform
        if (grouped && tid == 0) L.loss = L.loss + gloss;
    }
    // Short atoms (a few elements per lane): the whole group by ONE wave, atom after atom, without a workgroup barrier per
    // atom.  Lane l carries the four strided partial sums l, 64+l, 128+l, 192+l of the workgroup form (wave_window_energy):
    // the same trees, the same bits.
    const bool wave_apply = Pol::kWaveApply && W * F <= 1024;            // uniform
    if (applied_ahead) sy.full();                                        // (the owning wave has stored the span; its loss is in L.loss)
    else if (grouped) sy.full();
    else if (wave_apply) {
        if (wv == 0)
            for (int gi = 0; gi < n; ++gi) {
                const int tp = T_(gi), kk = K_(gi);
                const R cf = A_(gi);
                int s, e, es;
                const int cnt = centered_span(T, W, tp, s, e, es) * F;
                const R nc = -cf;
                const R* dk = S.D + ((int64_t)kk * W + es) * F;
                R* rv = G.r + (int64_t)s * F;
                R b4[4] = {(R)0, (R)0, (R)0, (R)0}, a4[4] = {(R)0, (R)0, (R)0, (R)0};
                for (int i0 = lane; i0 < cnt; i0 += kThreads) {
                    R v[4], d[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u; v[u] = (R)0; d[u] = (R)0; if (i < cnt) { v[u] = rv[i]; d[u] = dk[i]; } }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + 64 * u;
                        if (i < cnt) {
                            const R sq = v[u] * v[u];
                            b4[u] = b4[u] + sq;
                            const R prod = nc * d[u];            // -c*D[k] rounded, then += (utils.py:120,129)
                            const R vn = v[u] + prod;
                            rv[i] = vn;
                            const R sq2 = vn * vn;
                            a4[u] = a4[u] + sq2;
                        }
                    }
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const R ob = __shfl_down(b4[u], m), oa = __shfl_down(a4[u], m);
                        b4[u] = b4[u] + ob; a4[u] = a4[u] + oa;
                    }
                }
                if (lane == 0) {
                    const R b01 = b4[0] + b4[1], b23 = b4[2] + b4[3], a01 = a4[0] + a4[1], a23 = a4[2] + a4[3];
                    const R pb = b01 + b23, pa = a01 + a23;
                    const R l = pb - pa;
                    L.loss = L.loss + l;                         // :1005 summed over the group
                }
                __threadfence_block();                           // this atom's residual stores before the next atom's loads
            }
        sy.full();
    } else
    for (int gi = 0; gi < n; ++gi) {
        const int tp = T_(gi), kk = K_(gi);
        const R cf = A_(gi);
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
    HSCMP_STAMP(4);                                                      // coefficients, subtractions, local energies
    if (tid == 0) sh.e_res = sh.e_res - L.loss;                          // :1014
    if (P.has_scale)
        for (int gi = 0; gi < n; ++gi) {
            int s, e, es;
            centered_span(T, W, T_(gi), s, e, es);
            const int sg0 = s >> P.seg_shift, sg1 = (e - 1) >> P.seg_shift;
            for (int sg = sg0 + wv; sg <= sg1; sg += kWaves) rscan_segment(P, G, sh, sg, lane);
        }

    // ---- :1353 re-correlation around every atom of the group (from the final residual), then the maxima of their segments
    Pol::before_runs(A, plds);
    sy.full();
    bool joint = false;
    int pmin = p, pmax = p;
    if constexpr (Pol::kUnionRows) { joint = (n > 1 || Pol::kLoneRows) && L.jout == 0; pmin = L.jmin; pmax = L.jmax; }       // uniform
    bool deferred = false;
    if (joint && pmax - pmin <= 2 * W) {
        if (may_defer && owner >= 0) {                                   // uniform: the owning wave does them behind the batch's last selection
            if (wv == owner) { rows.pending = 1; rows.pmin = pmin; rows.pmax = pmax; }
            deferred = true;
        } else {
            Pol::lrun_span(P, S, G, A, plds, pmin, pmax, sy);
            sy.full();
        }
    } else
    for (int gi = 0; gi < n; ++gi) {
        Pol::lrun(P, S, G, sh, A, plds, T_(gi), K_(gi), sy);
        sy.full();
    }
    HSCMP_STAMP(5);                                                      // re-correlation
    if (P.blocked) {                                                     // the round end rescans the marked segments: one thread per atom
        for (int gi = tid; gi < n; gi += kThreads) {
            const int tp = T_(gi);
            const int sg0 = max(0, tp - (W - 1)) >> P.seg_shift, sg1 = min(T - 1, tp + (W - 1)) >> P.seg_shift;
            for (int sg = sg0; sg <= sg1; ++sg) atomicOr(&sh.touched[sg >> 5], 1u << (sg & 31));
        }
    } else
    for (int gi = 0; gi < n; ++gi) {
        const int tp = T_(gi);
        const int lo = max(0, tp - (W - 1)), hi = min(T - 1, tp + (W - 1));
        const int sg0 = lo >> P.seg_shift, sg1 = hi >> P.seg_shift;
        for (int sg = sg0 + wv; sg <= sg1; sg += kWaves) scan_segment<false>(P, G, wts, sh, sg, lane);
    }

    // ---- :1357-1383 fast stop rules
    HSCMP_STAMP(6);                                                      // segments marked / rescanned
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
    return deferred ? 1 : 0;
}

// the rows that waited (locomp_atom, may_defer): every wave those of the selection it owns, side by side; all threads of the workgroup
template <typename R, typename Pol, typename SY>
__device__ __forceinline__ void locomp_rows_deferred(const DevParams& P, const State<R>& S, const Sig<R>& G, const typename Pol::Args& A, char* plds,
                                                     LocompRows& rows, SY& sy)
{
    if (rows.pending) Pol::rows_of_wave(P, S, G, A, plds, rows.pmin, rows.pmax);
    rows.pending = 0;
    sy.full();
}

}  // namespace hscmp
