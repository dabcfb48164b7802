// hscmp_rp_sparse.h -- policy of the round-parallel loop (hscmp_rp.h) for multi-feature inputs with a sparse dictionary:
// the hierarchical levels >= 1 (modeling.py:1427-1492), where iterate_kernel<SparseRecorr> spends ~15 us per atom in a
// chain of dependent memory round trips with 256 threads mostly waiting.
//
// Here every piece of an atom is the work of ONE wave with a slot of LDS of its own -- the lists of a level-1 window are
// short (~27 non-zero cells, ~50 non-zero products on BASELINE config 5) -- and the atoms of a blocked round go through
// each phase side by side, one wave each:
//   candidate / energies  the atom's span gathered through the per-row feature lists (hscmp_sparse.h), the atom's
//                         non-zeros applied to the copy: local energies before / after (:1002-1005) in the pinned order
//                         (256 strided partial sums, at most two cells each -- otherwise, or when a row list has
//                         overflowed, the dense walk of the generic kernel runs, by the same wave)
//   subtract              the atom's non-zeros into the residual, its cells into the row lists
//   recorrelate           the 3W-2 window gathered from the final residual, paired with the by-feature lists of the
//                         dictionary (staged in LDS), sorted by (output row, atom, chain order), one pinned fma chain per
//                         listed output, per-row arg-max (hscmp_sparse.h::sparse_rows, restated for 64 lanes and
//                         wave-ordered LDS).  The slot is small (16 waves share the CU's LDS): a range of rows whose
//                         lists do not fit is halved and taken again; a single row that still does not fit walks its
//                         atoms' non-zeros.
// Same arithmetic, same order as iterate_kernel<SparseRecorr>: bit-identical results.
#pragma once

#include "hscmp_rp.h"
#include "hscmp_sparse.h"

namespace hscmp {

#ifdef HSCMP_DBG_STAMPS
// diagnostic build only: event counters of workgroup 0 (any wave), read through hscmp_debug_counters()
#define HSCMP_RP_TALLY(i, v) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_cnt[i], (unsigned long long)(v)); } while (0)
#else
#define HSCMP_RP_TALLY(i, v) do {} while (0)
#endif

#ifdef HSCMP_DBG_STAMPS
// diagnostic build only: cycle sums of the stages of wave 0 of workgroup 0 (g_stamps[32..])
#define HSCMP_RP_SSTAMP_BEGIN() unsigned long long sst_last_ = clock64()
#define HSCMP_RP_SSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_stamps[i] += now_ - sst_last_; sst_last_ = now_; } } while (0)
#else
#define HSCMP_RP_SSTAMP_BEGIN() do {} while (0)
#define HSCMP_RP_SSTAMP(i) do {} while (0)
#endif

struct RpSparseCaps {
    int nz;          // gathered non-zero cells of a window (or of a span, for the energies)
    int rec;         // non-zero products of a row range
    int teams;       // slots = waves that take per-atom work
    int stage_lists; // the by-feature lists of the dictionary are copied to LDS
    int stage_atoms; // the per-atom lists of the dictionary are copied to LDS
};

template <typename R> struct RpSparseArgs {
    SparseArgs<R> sp;      // dictionary lists, row lists (hscmp_sparse.h)
    RpSparseCaps caps;
};

// ---- what the waves share: weights, per-atom list offsets, by-feature lists, per-atom lists (as much as fits) -----
template <typename R> __host__ __device__ inline size_t rp_weight_bytes(const DevParams& P, const SparseArgs<R>& A)
{
    return A.wts ? ((size_t)P.K * sizeof(R) + 15) / 16 * 16 : 0;
}
template <typename R> __host__ __device__ inline size_t rp_ptr_bytes(const DevParams& P) { return (((size_t)P.K + 1) * 4 + 15) / 16 * 16; }
template <typename R> __host__ __device__ inline size_t rp_flist_bytes(const DevParams& P, const SparseArgs<R>& A)
{
    return ((((size_t)(P.F + 1) + (size_t)A.nnz) * 4 + 7) / 8 * 8 + (size_t)A.nnz * sizeof(R) + 15) / 16 * 16;
}
template <typename R> __host__ __device__ inline size_t rp_alist_bytes(const SparseArgs<R>& A)
{
    return (((size_t)A.nnz * 4 + 7) / 8 * 8 + (size_t)A.nnz * sizeof(R) + 15) / 16 * 16;
}
template <typename R> __host__ __device__ inline size_t rp_shared_dict_bytes(const DevParams& P, const RpSparseArgs<R>& A)
{
    return rp_weight_bytes(P, A.sp) + rp_ptr_bytes<R>(P) + (A.caps.stage_lists ? rp_flist_bytes(P, A.sp) : 0) + (A.caps.stage_atoms ? rp_alist_bytes(A.sp) : 0);
}
template <typename R> __device__ __forceinline__ SparseArgs<R> rp_dict_view(const DevParams& P, const RpSparseArgs<R>& A, char* base)
{
    SparseArgs<R> B = A.sp;
    char* p = base;
    if (A.sp.wts) { B.wts = reinterpret_cast<const R*>(p); p += rp_weight_bytes(P, A.sp); }
    B.nzptr = reinterpret_cast<const int*>(p); p += rp_ptr_bytes<R>(P);
    if (A.caps.stage_lists) {
        int* fptr = reinterpret_cast<int*>(p);
        B.fptr = fptr; B.fkw = fptr + (P.F + 1);
        B.fval = reinterpret_cast<const R*>(p + (((size_t)(P.F + 1) + (size_t)A.sp.nnz) * 4 + 7) / 8 * 8);
        p += rp_flist_bytes(P, A.sp);
    }
    if (A.caps.stage_atoms) {
        B.nzwf = reinterpret_cast<const int*>(p);
        B.nzval = reinterpret_cast<const R*>(p + ((size_t)A.sp.nnz * 4 + 7) / 8 * 8);
    }
    return B;
}

// ---- a wave's slot -------------------------------------------------------------------------------------------------
template <typename R> __host__ __device__ inline size_t rp_sparse_slot_bytes(const RpSparseCaps& c)
{
    // re-correlation: records (key 8, two factors, chain result, perm, okey), non-zero list (value, key)
    const size_t rc = 16 + (size_t)c.rec * (8 + 3 * sizeof(R) + 4 + 4) + (size_t)c.nz * (sizeof(R) + 4);
    // energies: 256 partial sums (count, four cell numbers), cell list (key, before, after)
    const size_t en = 16 + 256 * (4 + 8) + (size_t)c.nz * (4 + 2 * sizeof(R));
    return ((rc > en ? rc : en) + 15) / 16 * 16;
}

template <typename R> struct RpSparseSlot {
    int* ctl;                    // [4]
    unsigned long long* rkey;    // [rec]
    R* rx; R* rd; R* out;        // [rec]
    int* perm; unsigned* okey;   // [rec]
    R* val; int* key;            // [nz]
    // energies (over the same bytes): per partial sum its cell count and two cell numbers; the cells
    int* members; unsigned short* cell01; int* ckey; R* cbefore; R* cafter;
    // per-row results (alias the record arrays once the chains have run; rows of a range <= rec)
    unsigned long long* rmax; int* rk; R* rc; R* c0;
};

template <typename R> __device__ __forceinline__ RpSparseSlot<R> rp_sparse_slot(char* base, const RpSparseCaps& c)
{
    RpSparseSlot<R> L;
    L.ctl = reinterpret_cast<int*>(base);
    char* p = base + 16;
    L.rkey = reinterpret_cast<unsigned long long*>(p); p += (size_t)c.rec * 8;
    L.rx = reinterpret_cast<R*>(p); p += (size_t)c.rec * sizeof(R);
    L.rd = reinterpret_cast<R*>(p); p += (size_t)c.rec * sizeof(R);
    L.out = reinterpret_cast<R*>(p); p += (size_t)c.rec * sizeof(R);
    L.val = reinterpret_cast<R*>(p); p += (size_t)c.nz * sizeof(R);
    L.perm = reinterpret_cast<int*>(p); p += (size_t)c.rec * 4;
    L.okey = reinterpret_cast<unsigned*>(p); p += (size_t)c.rec * 4;
    L.key = reinterpret_cast<int*>(p);
    char* q = base + 16;
    L.cbefore = reinterpret_cast<R*>(q); q += (size_t)c.nz * sizeof(R);
    L.cafter = reinterpret_cast<R*>(q); q += (size_t)c.nz * sizeof(R);
    L.ckey = reinterpret_cast<int*>(q); q += (size_t)c.nz * 4;
    L.members = reinterpret_cast<int*>(q); q += 256 * 4;
    L.cell01 = reinterpret_cast<unsigned short*>(q);
    L.rmax = L.rkey; L.rc = L.rd; L.c0 = L.rx; L.rk = L.perm;
    return L;
}

template <typename R> struct RpSparse {
    static constexpr int kMaxSegments = 512;
    static constexpr int kMaxSel = 64;
    static constexpr bool kScoreOnly = false;
    using Shared = RpShared<R, kMaxSegments, kMaxSel>;
    using Args = RpSparseArgs<R>;

    static __host__ __device__ size_t policy_lds_bytes(const DevParams& P, const Args& A)
    {
        return rp_shared_dict_bytes(P, A) + (size_t)A.caps.teams * rp_sparse_slot_bytes<R>(A.caps);
    }
    static size_t total_lds_bytes(const DevParams& P, const Args& A) { return ((sizeof(Shared) + 15) / 16) * 16 + policy_lds_bytes(P, A); }
    static __device__ __forceinline__ char* slot_base(const DevParams& P, const Args& A, char* lds, int wv)
    {
        return lds + rp_shared_dict_bytes(P, A) + (size_t)wv * rp_sparse_slot_bytes<R>(A.caps);
    }
    static __device__ __forceinline__ const R* weights(const DevParams& P, const State<R>&, const Args& A, char* lds) { return rp_dict_view(P, A, lds).wts; }
    static __device__ __forceinline__ int units_per_atom(const DevParams&) { return 1; }
    static __device__ __forceinline__ int teams(const Args& A) { return A.caps.teams; }
    static __device__ __forceinline__ void after_atom(const DevParams&, const Args&, char*, int) {}
    static __device__ __forceinline__ void epilogue(const DevParams&, const State<R>&, const Args&, char*, int) {}

    static __device__ __forceinline__ void prologue(const DevParams& P, const State<R>&, const Sig<R>&, const Args& A, char* lds, int)
    {
        const SparseArgs<R> B = rp_dict_view(P, A, lds);
        const int tid = threadIdx.x;
        if (A.sp.wts) for (int i = tid; i < P.K; i += kRpThreads) const_cast<R*>(B.wts)[i] = A.sp.wts[i];
        for (int i = tid; i <= P.K; i += kRpThreads) const_cast<int*>(B.nzptr)[i] = A.sp.nzptr[i];
        if (A.caps.stage_lists) {
            for (int i = tid; i <= P.F; i += kRpThreads) const_cast<int*>(B.fptr)[i] = A.sp.fptr[i];
            for (int i = tid; i < A.sp.nnz; i += kRpThreads) { const_cast<int*>(B.fkw)[i] = A.sp.fkw[i]; const_cast<R*>(B.fval)[i] = A.sp.fval[i]; }
        }
        if (A.caps.stage_atoms)
            for (int i = tid; i < A.sp.nnz; i += kRpThreads) { const_cast<int*>(B.nzwf)[i] = A.sp.nzwf[i]; const_cast<R*>(B.nzval)[i] = A.sp.nzval[i]; }
        __syncthreads();
    }

    static __device__ __forceinline__ void fence()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    static __device__ __forceinline__ bool is_interior(const DevParams& P, int p)
    {
        const int tstart = p - P.off - (P.W - 1), tend = p + P.W / 2 + (P.W - 1);
        return tstart >= 0 && tend <= P.T - 1;
    }

    // ---- local energies of the atom at t (:1002-1005): the listed cells of its clipped span, the atom's non-zeros applied
    //      to the copy.  The pinned order is 256 strided partial sums, each sequential in the cell index: every cell finds its
    //      rank among the cells of its partial sum (by a scan of the cell list: a dozen entries), and the lane of a partial sum
    //      adds its (up to four) cells in that order.  A partial sum with more cells, a row list of the span that has
    //      overflowed, or more cells than the slot holds: the dense walk (the generic kernel's own loop) runs instead.
    //      One wave; result in every lane.  k < 0: (k, c) of the position are fetched here, beside the row lists.
    static __device__ __forceinline__ void span_energies(const DevParams& P, const State<R>& S, const Sig<R>& G, const SparseArgs<R>& A,
                                                         const RpSparseSlot<R>& L, int nzcap, int t, int& k, R& c, int lane, R& eb_out, R& ea_out)
    {
        const int T = P.T, F = P.F, W = P.W;
        int s, e, es;
        const int len = centered_span(T, W, t, s, e, es);
        const int* cnt = A.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A.rl_f + (int64_t)blockIdx.x * T * 8;
        HSCMP_RP_SSTAMP_BEGIN();
        fence();                                              // (the slot's previous user -- this wave -- is done with it)
#pragma unroll
        for (int u = 0; u < 4; ++u) L.members[lane + 64 * u] = 0;
        if (lane == 0) L.ctl[0] = 0;
        bool bad = false;
        // the span's listed cells: a row's count and list in one round trip (with the position's atom), its cells in the next
        constexpr int kPass = 2;                              // rows of the span per lane (W <= 128)
        int n_[kPass]; int4 a_[kPass], b_[kPass];
#pragma unroll
        for (int ps = 0; ps < kPass; ++ps) {
            const int g = s + lane + 64 * ps;
            const int gq = g < e ? g : s;
            n_[ps] = list_count(cnt + gq);
            const int4* row = reinterpret_cast<const int4*>(lf + (int64_t)gq * 8);
            a_[ps] = row[0]; b_[ps] = row[1];
        }
        if (k < 0) { k = __builtin_amdgcn_readfirstlane(G.bk[t]); c = wave_bcast(G.bc[t], 0); }
        const R nc = -c;
        fence();
        HSCMP_RP_SSTAMP(32);
#pragma unroll
        for (int ps = 0; ps < kPass; ++ps) {
            if (s + 64 * ps >= e) break;                      // (uniform: a span of at most 64 rows has no second pass)
            const int g = s + lane + 64 * ps;
            const bool row_on = g < e;
            const int fs[8] = {a_[ps].x, a_[ps].y, a_[ps].z, a_[ps].w, b_[ps].x, b_[ps].y, b_[ps].z, b_[ps].w};
            const bool cells_on = row_on && n_[ps] > 0 && n_[ps] <= 8;
            R vs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vs[u] = (cells_on && fs[u] >= 0) ? G.r[(int64_t)g * F + fs[u]] : (R)0;
            // (a lane reserves room for all its non-zero cells with ONE counter update)
            int mine = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) mine += (vs[u] != (R)0) ? 1 : 0;
            int j = mine > 0 ? atomicAdd(&L.ctl[0], mine) : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (vs[u] != (R)0) {
                    if (j < nzcap) { L.ckey[j] = (g - s) * F + fs[u]; L.cbefore[j] = vs[u]; L.cafter[j] = vs[u]; }
                    ++j;
                }
            // rows whose list has overflowed (cancelled cells stay listed: a busy row fills up) are read densely -- those
            // rows only, the lanes over the features
            for (unsigned long long ovf = __ballot(row_on && n_[ps] > 8); ovf; ovf &= ovf - 1ull) {          // (uniform)
                const int go = s + 64 * ps + (__ffsll((long long)ovf) - 1);
                for (int f0 = 0; f0 < F; f0 += 256) {
                    R v4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int f = f0 + lane + 64 * u; v4[u] = f < F ? G.r[(int64_t)go * F + f] : (R)0; }
                    int cntz = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) cntz += (v4[u] != (R)0) ? 1 : 0;
                    int jo = cntz > 0 ? atomicAdd(&L.ctl[0], cntz) : 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (v4[u] != (R)0) {
                            if (jo < nzcap) { L.ckey[jo] = (go - s) * F + f0 + lane + 64 * u; L.cbefore[jo] = v4[u]; L.cafter[jo] = v4[u]; }
                            ++jo;
                        }
                }
            }
        }
        fence();
        for (int g0 = s + 64 * kPass; g0 < e; g0 += 64) bad = true;           // (W > 128: not on this path)
        HSCMP_RP_SSTAMP(33);
        const int n0 = min(L.ctl[0], nzcap);
        if (L.ctl[0] > nzcap) bad = true;
        // the atom's non-zeros: -c*D[k] rounded, then += (utils.py:120,129); a cell that is not listed yet starts from 0
        const int e0 = A.nzptr[k], e1 = A.nzptr[k + 1];
        for (int q0 = e0; q0 < e1; q0 += 64) {
            const int qe = q0 + lane;
            bool fresh = false;
            int i = 0;
            R prod = (R)0;
            if (qe < e1) {
                const int wf = A.nzwf[qe], f = wf & 0xffff, g = t - P.off + (wf >> 16);
                if (g >= s && g < e) {                        // (the clipped part of the atom touches nothing)
                    prod = nc * A.nzval[qe];
                    i = (g - s) * F + f;
                    int hit = n0;
                    for (int j0 = 0; j0 < n0 && hit == n0; j0 += 4) {       // four list entries per LDS round trip
                        const int k0 = L.ckey[j0], k1 = j0 + 1 < n0 ? L.ckey[j0 + 1] : -1, k2 = j0 + 2 < n0 ? L.ckey[j0 + 2] : -1, k3 = j0 + 3 < n0 ? L.ckey[j0 + 3] : -1;
                        hit = k0 == i ? j0 : k1 == i ? j0 + 1 : k2 == i ? j0 + 2 : k3 == i ? j0 + 3 : n0;
                    }
                    if (hit < n0) L.cafter[hit] = L.cbefore[hit] + prod;
                    else fresh = true;
                }
            }
            const unsigned long long mf = __ballot(fresh);
            if (mf) {
                const int base = L.ctl[0];
                const int j = base + __popcll(mf & ((1ull << lane) - 1ull));
                if (fresh && j < nzcap) { L.ckey[j] = i; L.cbefore[j] = (R)0; L.cafter[j] = (R)0 + prod; }
                fence();
                if (lane == 0) L.ctl[0] = base + __popcll(mf);
                fence();
            }
        }
        fence();
        const int n = L.ctl[0];
        if (n > nzcap) bad = true;
        HSCMP_RP_SSTAMP(34);
        // The pinned order: partial sum q adds the cells with index = q mod 256 in ascending index.  Up to 64 cells (the
        // rule): every lane holds one; the cells are sorted by (partial sum, index) with a rank count over v_readlane
        // broadcasts, their squares go to LDS in that order and the lane of a partial sum walks its run, however long.
        // More cells than lanes: the dense walk below.
        if (n > 64) bad = true;
        R pb[4] = {(R)0, (R)0, (R)0, (R)0}, pa[4] = {(R)0, (R)0, (R)0, (R)0};
        if (__ballot(bad) == 0ull) {
            const int i = lane < n ? L.ckey[lane] : -1;
            const R vb = lane < n ? L.cbefore[lane] : (R)0, va = lane < n ? L.cafter[lane] : (R)0;
            fence();                                           // (the squares below go over the cell list)
            R* sb = L.cbefore; R* sa = L.cafter;               // [64] squares before / after, sorted
            unsigned short* first = L.cell01;                  // [256] start of the run of a partial sum
            const int skey = i >= 0 ? ((i & 255) << 22) | (i >> 8) : (0x40000000 | lane);
            int pos = 0, before_same = 0;
            for (int q = 0; q < n; ++q) {
                const int kq = __builtin_amdgcn_readlane(skey, q);
                pos += kq < skey ? 1 : 0;
                before_same += (kq < skey && (kq >> 22) == (skey >> 22)) ? 1 : 0;
            }
            if (i >= 0) {
                sb[pos] = vb * vb; sa[pos] = va * va;
                atomicAdd(&L.members[i & 255], 1);
                if (before_same == 0) first[i & 255] = (unsigned short)pos;
            }
            fence();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = lane + 64 * u;
                const int m = L.members[q];
                const int f0 = m > 0 ? first[q] : 0;
                R bsum = (R)0, asum = (R)0;
                for (int r = 0; r < m; ++r) { bsum = bsum + sb[f0 + r]; asum = asum + sa[f0 + r]; }
                pb[u] = bsum; pa[u] = asum;
            }
        }
        HSCMP_RP_SSTAMP(35);
        HSCMP_RP_TALLY(0, 1); HSCMP_RP_TALLY(1, __ballot(bad) != 0ull); HSCMP_RP_TALLY(2, L.ctl[0]);
        if (__ballot(bad) != 0ull) {
            // dense walk: partial sum q = cell index mod 256, ascending (modeling.py:996-1016 as the generic kernel runs it)
            const int nn = len * F;
            const R* dk = S.D + ((int64_t)k * W + es) * F;
            const R* rv = G.r + (int64_t)s * F;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                R b = (R)0, a = (R)0;
                for (int i = lane + 64 * u; i < nn; i += 256) {
                    const R v = rv[i];
                    const R prod = nc * dk[i];
                    const R vn = v + prod;
                    const R sq = v * v, sq2 = vn * vn;
                    b = b + sq; a = a + sq2;
                }
                pb[u] = b; pa[u] = a;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) rp_tree_down2(pb[u], pa[u]);
        const R b01 = pb[0] + pb[1], b23 = pb[2] + pb[3], a01 = pa[0] + pa[1], a23 = pa[2] + pa[3];
        eb_out = wave_bcast(b01 + b23, 0);
        ea_out = wave_bcast(a01 + a23, 0);
        fence();
        HSCMP_RP_SSTAMP(36);
    }

    static __device__ __forceinline__ void candidate(const DevParams& P, const State<R>& S, const Sig<R>& G, const Args& A0, char* lds,
                                                     int t, int lane, int wv, int& k_out, R& c_out, R& eb, R& ea, int& flag)
    {
        const SparseArgs<R> A = rp_dict_view(P, A0, lds);
        const RpSparseSlot<R> L = rp_sparse_slot<R>(slot_base(P, A0, lds, wv), A0.caps);
        k_out = -1; c_out = (R)0;
        span_energies(P, S, G, A, L, A0.caps.nz, t, k_out, c_out, lane, eb, ea);
        flag = is_interior(P, t) ? RPF_INTERIOR : 0;
    }

    static __device__ __forceinline__ void energies(const DevParams& P, const State<R>& S, const Sig<R>& G, const Args& A0, char* lds,
                                                    int t, int k, R c, int lane, int wv, R& eb, R& ea)
    {
        const SparseArgs<R> A = rp_dict_view(P, A0, lds);
        const RpSparseSlot<R> L = rp_sparse_slot<R>(slot_base(P, A0, lds, wv), A0.caps);
        span_energies(P, S, G, A, L, A0.caps.nz, t, k, c, lane, eb, ea);
    }

    // ---- the atom's non-zeros into the residual (:996-1016 restricted to them: a zero of the atom changes nothing), its
    //      cells into the row lists (the window gathers read the lists instead of scanning rows of F values)
    static __device__ __forceinline__ void subtract(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A0, char* lds,
                                                    int p, int k, R c, int lane, int)
    {
        const SparseArgs<R> A = rp_dict_view(P, A0, lds);
        const int T = P.T, F = P.F;
        int* cntw = A.rl_cnt + (int64_t)blockIdx.x * T;
        int* lfw = A.rl_f + (int64_t)blockIdx.x * T * 8;
        const R nc = -c;
        const int e0 = A.nzptr[k], e1 = A.nzptr[k + 1];
        for (int q0 = e0; q0 < e1; q0 += 64) {
            const int qe = q0 + lane;
            if (qe >= e1) continue;
            const int wf = A.nzwf[qe], f = wf & 0xffff, g = p - P.off + (wf >> 16);
            if (g < 0 || g >= T) continue;                    // clipped part of the atom (utils.py:110-129)
            const R prod = nc * A.nzval[qe];
            R* cell = G.r + (int64_t)g * F + f;
            const R v = *cell;
            const int n2 = list_count(cntw + g);
            const int4* row2 = reinterpret_cast<const int4*>(lfw + (int64_t)g * 8);
            const int4 a2 = row2[0], b2 = row2[1];
            *cell = v + prod;
            // (empty slots hold -1 and never match; an overflowed row, count > 8, is read densely anyway)
            const bool listed = n2 > 8 || a2.x == f || a2.y == f || a2.z == f || a2.w == f || b2.x == f || b2.y == f || b2.z == f || b2.w == f;
            if (!listed) {
                const int o = atomicAdd(&cntw[g], 1);
                if (o < 8) lfw[(int64_t)g * 8 + o] = f;
            }
        }
    }

    // rows [r0, r0 + nr) of the atom's touched rows (row 0 = position p-(W-1)) through the listed cells of their window.
    // false (nothing written): a row list of the window has overflowed, or the cells / products do not fit the slot.
    static __device__ __forceinline__ bool rows_listed(const DevParams& P, const Sig<R>& G, const SparseArgs<R>& A, const RpSparseSlot<R>& L,
                                                       const RpSparseCaps& caps, int p, int r0, int nr, bool interior, int sidx, int nslice, int lane)
    {
        const int T = P.T, F = P.F, W = P.W;
        const int row0 = p - (W - 1) + r0, nwin = nr + W - 1, g0 = row0 - P.off;
        const int* cnt = A.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A.rl_f + (int64_t)blockIdx.x * T * 8;
        HSCMP_RP_SSTAMP_BEGIN();
        fence();
        if (lane < 4) L.ctl[lane] = 0;
        fence();
        // 1. the non-zero cells of the window (any order): [0] count, [1] overflowed rows.  Window rows in passes of 64, the
        //    loads of two passes in flight together.
        for (int j0 = 0; j0 < nwin; j0 += 128) {
            int n_[2], gg_[2]; int4 a_[2], b_[2];
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int j = j0 + 64 * ps + lane;
                const int jq = j < nwin ? j : 0;
                gg_[ps] = interior ? g0 + jq : reflect_index(g0 + jq, sidx, nslice);        // np.pad 'reflect', :1046
                n_[ps] = list_count(cnt + gg_[ps]);
                const int4* row = reinterpret_cast<const int4*>(lf + (int64_t)gg_[ps] * 8);
                a_[ps] = row[0]; b_[ps] = row[1];
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int j = j0 + 64 * ps + lane;
                const bool on = j < nwin;
                // a row whose list has overflowed is read densely (that row only), the lanes over the features
                for (unsigned long long ovf = __ballot(on && n_[ps] > 8); ovf; ovf &= ovf - 1ull) {              // (uniform)
                    const int src = __ffsll((long long)ovf) - 1;
                    const int jo = j0 + 64 * ps + src, go = __builtin_amdgcn_readlane(gg_[ps], src);
                    for (int f0 = 0; f0 < F; f0 += 256) {
                        R v4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int f = f0 + lane + 64 * u; v4[u] = f < F ? G.r[(int64_t)go * F + f] : (R)0; }
                        int cntz = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) cntz += (v4[u] != (R)0) ? 1 : 0;
                        int oo = cntz > 0 ? atomicAdd(&L.ctl[0], cntz) : 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (v4[u] != (R)0) {
                                if (oo < caps.nz) { L.val[oo] = v4[u]; L.key[oo] = ((f0 + lane + 64 * u) << 16) | jo; }
                                ++oo;
                            }
                    }
                }
                const bool cells_on = on && n_[ps] > 0 && n_[ps] <= 8;
                const int fs[8] = {a_[ps].x, a_[ps].y, a_[ps].z, a_[ps].w, b_[ps].x, b_[ps].y, b_[ps].z, b_[ps].w};
                R vs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vs[u] = (cells_on && fs[u] >= 0) ? G.r[(int64_t)gg_[ps] * F + fs[u]] : (R)0;
                int mine = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) mine += (vs[u] != (R)0) ? 1 : 0;
                int o = mine > 0 ? atomicAdd(&L.ctl[0], mine) : 0;          // (room for all the lane's cells with one update)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (vs[u] != (R)0) {
                        if (o < caps.nz) { L.val[o] = vs[u]; L.key[o] = (fs[u] << 16) | j; }
                        ++o;
                    }
            }
        }
        fence();
        const int n = L.ctl[0];
        HSCMP_RP_TALLY(3, 1); HSCMP_RP_TALLY(4, L.ctl[1] != 0); HSCMP_RP_TALLY(5, n > caps.nz); HSCMP_RP_TALLY(6, n);
        HSCMP_RP_SSTAMP(40);
        if (L.ctl[1] != 0 || n > caps.nz) return false;
        // 2. every input non-zero (f, j) with the dictionary non-zeros (k, w) of feature f: output row j - w
        int longest = 0;
        for (int i = lane; i < n; i += 64) {
            const int f = L.key[i] >> 16;
            longest = max(longest, A.fptr[f + 1] - A.fptr[f]);
        }
        longest = wave_max_i32(longest);
        const float inv = longest > 0 ? 1.0f / (float)longest : 0.0f;
        int m = 0;                                                           // (uniform: the records are appended in lane order)
        for (int it0 = 0; it0 < n * longest; it0 += 64) {
            const int it = it0 + lane;
            bool on = it < n * longest;
            int i = 0, f = 0, kw = 0, row = 0;
            R d = (R)0, x = (R)0;
            if (on) {
                i = (int)(((float)it + 0.5f) * inv);                         // it / longest (exact: it < 2^17)
                int sl = it - i * longest;
                if (sl < 0) { --i; sl += longest; } else if (sl >= longest) { ++i; sl -= longest; }
                const int key = L.key[i], j = key & 0xffff;
                f = key >> 16;
                const int b = A.fptr[f], len = A.fptr[f + 1] - b;
                on = sl < len;
                if (on) {
                    kw = A.fkw[b + sl]; d = A.fval[b + sl]; x = L.val[i];
                    row = j - (kw & 0xffff);
                    const int t = row0 + row;
                    on = row >= 0 && row < nr && t >= 0 && t < T;
                }
            }
            const unsigned long long mk = __ballot(on);
            const int o = m + __popcll(mk & ((1ull << lane) - 1ull));
            if (on && o < caps.rec) {
                L.rkey[o] = ((unsigned long long)row << 48) | ((unsigned long long)((unsigned)kw >> 16) << 32) |
                            ((unsigned long long)f << 16) | (unsigned)(kw & 0xffff);
                L.rx[o] = x; L.rd[o] = d;
            }
            m += __popcll(mk);
        }
        fence();
        HSCMP_RP_TALLY(7, m > caps.rec); HSCMP_RP_TALLY(8, m); HSCMP_RP_TALLY(10, longest);
        HSCMP_RP_SSTAMP(41);
        if (m > caps.rec) return false;
        // 3. sort by (output, chain order); the keys are distinct.  Up to 64 records: every lane keeps its key and counts the
        //    smaller ones among v_readlane broadcasts; more: rank sort over the list, eight keys per LDS round trip.
        if (m <= 64) {
            const unsigned long long key = lane < m ? L.rkey[lane] : ~0ull;
            const int klo = (int)(unsigned)key, khi = (int)(unsigned)(key >> 32);
            int rank = 0;
            for (int q = 0; q < m; ++q) {
                const unsigned long long kq = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(khi, q) << 32) | (unsigned)__builtin_amdgcn_readlane(klo, q);
                rank += (kq < key) ? 1 : 0;
            }
            if (lane < m) L.perm[rank] = lane;
        } else
        for (int i = lane; i < m; i += 64) {
            const unsigned long long key = L.rkey[i];
            int rank = 0;
            int q = 0;
            for (; q + 8 <= m; q += 8) {
                unsigned long long kq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kq[u] = L.rkey[q + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += (kq[u] < key) ? 1 : 0;
            }
            for (; q < m; ++q) rank += (L.rkey[q] < key) ? 1 : 0;
            L.perm[rank] = i;
        }
        fence();
        HSCMP_RP_SSTAMP(42);
        // 4. one chain per output, run by the lane of its first record (f outer, w inner, from +0)
        for (int sp = lane; sp < m; sp += 64) {
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
        fence();
        HSCMP_RP_SSTAMP(43);
        // 5. per-row best over atoms: the listed outputs against the zeros of all the others -- a zero score never
        //    beats k = 0, the first of the ties; among equal scores the lowest atom wins (np.argmax)
        for (int row = lane; row < nr; row += 64) { L.rmax[row] = 0ull; L.rk[row] = INT_MAX; L.c0[row] = (R)0; }
        fence();
        for (int sp = lane; sp < m; sp += 64) {
            const unsigned ok = L.okey[sp];
            if (ok == ~0u) continue;
            const int row = (int)(ok >> 16), kk = (int)(ok & 0xffffu);
            const R o = L.out[sp];
            if (kk == 0) L.c0[row] = o;                                      // the value of the default winner
            const R sc = score_of(o, kk, A.wts);
            if (sc > (R)0) atomicMax(&L.rmax[row], score_bits(sc));
        }
        fence();
        for (int sp = lane; sp < m; sp += 64) {
            const unsigned ok = L.okey[sp];
            if (ok == ~0u) continue;
            const int row = (int)(ok >> 16), kk = (int)(ok & 0xffffu);
            const R sc = score_of(L.out[sp], kk, A.wts);
            if (sc > (R)0 && score_bits(sc) == L.rmax[row]) atomicMin(&L.rk[row], kk);
        }
        fence();
        for (int sp = lane; sp < m; sp += 64) {
            const unsigned ok = L.okey[sp];
            if (ok == ~0u) continue;
            const int row = (int)(ok >> 16), kk = (int)(ok & 0xffffu);
            const R o = L.out[sp];
            const R sc = score_of(o, kk, A.wts);
            if (sc > (R)0 && score_bits(sc) == L.rmax[row] && kk == L.rk[row]) L.rc[row] = o;
        }
        fence();
        for (int row = lane; row < nr; row += 64) {
            const int t = row0 + row;
            if (t < 0 || t >= T) continue;                                   // overlapReplace clipping (utils.py:133-161)
            const int k = L.rk[row];
            if (k == INT_MAX) { G.bc[t] = L.c0[row]; G.bk[t] = 0; }
            else { G.bc[t] = L.rc[row]; G.bk[t] = k; }
        }
        fence();
        HSCMP_RP_SSTAMP(44);
        return true;
    }

    // one row whose lists do not fit the slot: each output walks its atom's non-zeros (already in chain order), the lanes
    // over the atoms
    static __device__ __forceinline__ void row_by_atoms(const DevParams& P, const Sig<R>& G, const SparseArgs<R>& A, int p, int row,
                                                        bool interior, int sidx, int nslice, int lane)
    {
        const int T = P.T, F = P.F, W = P.W, K = P.K;
        const int t = p - (W - 1) + row, g0 = t - P.off;
        if (t < 0 || t >= T) return;                                         // (uniform)
        HSCMP_RP_TALLY(9, 1);
        Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
        R bc = (R)0;
        for (int k = lane; k < K; k += 64) {
            R acc = (R)0;
            const int e1 = A.nzptr[k + 1];
            for (int e = A.nzptr[k]; e < e1; ++e) {
                const int wf = A.nzwf[e];
                const int g = interior ? g0 + (wf >> 16) : reflect_index(g0 + (wf >> 16), sidx, nslice);
                acc = rfma(G.r[(int64_t)g * F + (wf & 0xffff)], A.nzval[e], acc);
            }
            const R sc = score_of(acc, k, A.wts);
            if (k == 0) bc = acc;                                             // (kept when no score compares: all NaN)
            if (sc > best.s) { best.s = sc; best.i = k; bc = acc; }           // ascending k per lane: '>' keeps the first of equals
        }
        const Cand<R> win = rp_wave_argmax(best);
        const int owner = __ffsll((long long)__ballot(best.i == win.i)) - 1;
        const R wc = wave_bcast(bc, owner);
        // (win.i == INT_MAX: every score of the row is NaN -- a diverged pursuit; np.argmax gives atom 0)
        if (lane == 0) { G.bc[t] = wc; G.bk[t] = win.i == INT_MAX ? 0 : win.i; }
    }

    // ---- rows p-(W-1) .. p+(W-1) from the (final) residual, reflect padded (:1018-1051): sparse window x sparse dictionary
    static __device__ __forceinline__ void recorrelate(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A0, char* lds,
                                                       int p, int, int, bool interior, int lane, int wv)
    {
        const SparseArgs<R> A = rp_dict_view(P, A0, lds);
        const RpSparseSlot<R> L = rp_sparse_slot<R>(slot_base(P, A0, lds, wv), A0.caps);
        const int T = P.T, W = P.W;
        const int nrows = 2 * W - 1;
        const int tstart = p - P.off - (W - 1), tend = p + W / 2 + (W - 1);
        const int sidx = tstart < 0 ? 0 : tstart, eidx = tend > T - 1 ? T - 1 : tend, nslice = eidx - sidx + 1;
        // all rows at once when their lists fit the slot; a range that does not fit is halved and taken again
        int step = min(nrows, A0.caps.rec);
        for (int r0 = 0; r0 < nrows;) {
            const int nr = min(step, nrows - r0);
            if (rows_listed(P, G, A, L, A0.caps, p, r0, nr, interior, sidx, nslice, lane)) { r0 += nr; continue; }
            if (nr > 1) { step = (nr + 1) / 2; continue; }
            row_by_atoms(P, G, A, p, r0, interior, sidx, nslice, lane);
            r0 += 1;
        }
        fence();
    }
};

// host-side dispatch -----------------------------------------------------------------------------
// What goes where in the CU's LDS: the control block, the shared dictionary lists (weights and per-atom offsets always;
// the by-feature lists and the per-atom lists when they leave room), then one slot per wave that takes per-atom work --
// as many as the blocks of a round can keep busy.
template <typename R> inline RpSparseCaps rp_sparse_caps(const DevParams& P, const SparseArgs<R>& sp, size_t fixed_bytes)
{
    RpSparseCaps c;
    c.nz = 96; c.rec = 128;
    c.stage_lists = 0; c.stage_atoms = 0; c.teams = 0;
    RpSparseArgs<R> A; A.sp = sp; A.caps = c;
    const size_t budget = (size_t)156 * 1024 - fixed_bytes;           // (a margin: the kernel's own static bytes count too)
    const int want = std::min(kRpWaves, std::max(4, P.maxsel));
    auto teams_for = [&](const RpSparseCaps& cc) {
        RpSparseArgs<R> B; B.sp = sp; B.caps = cc;
        const size_t shared = rp_shared_dict_bytes(P, B);
        return shared >= budget ? 0 : (int)std::min<size_t>(kRpWaves, (budget - shared) / rp_sparse_slot_bytes<R>(cc));
    };
    RpSparseCaps t = c;
    t.stage_lists = 1; if (teams_for(t) >= want) c = t;
    t = c; t.stage_atoms = 1; if (teams_for(t) >= want) c = t;
    // spare room: larger lists (fewer halved ranges)
    for (;;) { t = c; t.rec += 32; t.nz += 16; if (t.rec > 512 || teams_for(t) < want) break; c = t; }
    c.teams = std::min(want, teams_for(c));
    return c;
}

template <typename R>
static int rp_sparse_launch(hipStream_t stream, const DevParams& P0, const State<R>& S, const SparseArgs<R>& sp, bool dry)
{
    using Pol = RpSparse<R>;
    if (!rp_params_ok(P0, Pol::kMaxSel)) return -1;
    // the row lists (8 features per row) and the per-atom / by-feature dictionary lists carry this policy
    if (!sp.rl_cnt || sp.rl_cap != 8 || !sp.nzptr || !sp.fptr || P0.W > 128 || P0.K > 65535 || P0.F > 32767) return -1;
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    RpSparseArgs<R> A;
    A.sp = sp;
    A.caps = rp_sparse_caps<R>(P, sp, ((sizeof(typename Pol::Shared) + 15) / 16) * 16);
    if (A.caps.teams < 2) return -1;
    const size_t lds = Pol::total_lds_bytes(P, A);
    if (lds > (size_t)158 * 1024) return -1;
    auto kern = iterate_rp_kernel<R, Pol>;
    if (set_dyn_lds((const void*)kern, lds) != hipSuccess) return -1;
    if (dry) return 0;
    if (getenv("HSCMP_DEBUG"))
        fprintf(stderr, "[hscmp] iterate_rp_kernel<sparse>: LDS %zu B, %d teams, lists nz %d rec %d, staged by-feature %d per-atom %d\n", lds,
                A.caps.teams, A.caps.nz, A.caps.rec, A.caps.stage_lists, A.caps.stage_atoms);
    hipLaunchKernelGGL(kern, dim3(P.B), dim3(kRpThreads), lds, stream, P, S, A);
    return 0;
}

}  // namespace hscmp
