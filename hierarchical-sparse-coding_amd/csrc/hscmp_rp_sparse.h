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
//                         dictionary, sorted by (output row, atom, chain order), one pinned fma chain per listed output,
//                         per-row arg-max (hscmp_sparse.h::sparse_rows, restated for 64 lanes and wave-ordered LDS);
//                         a window or pair list that does not fit the slot takes the per-atom chains instead
// Same arithmetic, same order as iterate_kernel<SparseRecorr>: bit-identical results.
#pragma once

#include "hscmp_rp.h"
#include "hscmp_sparse.h"

namespace hscmp {

struct RpSparseCaps {
    int nz;        // gathered non-zero cells of a window
    int rec;       // non-zero products of a window
    int rows;      // 2W-1, rounded up
    int teams;     // slots = waves that take per-atom work
};

template <typename R> struct RpSparseArgs {
    SparseArgs<R> sp;      // dictionary lists, row lists (hscmp_sparse.h)
    RpSparseCaps caps;
};

template <typename R> __host__ __device__ inline size_t rp_sparse_slot_bytes(const RpSparseCaps& c)
{
    // records: key 8 + two factors + chain result; perm / okey ints; non-zero list: value + key; energy table: 256 x 2 entries
    const size_t rec = (size_t)c.rec * (8 + 3 * sizeof(R) + 4 + 4);
    const size_t nz = (size_t)c.nz * (sizeof(R) + 4);
    size_t bytes = 16 + rec + nz;
    const size_t etab = 16 + 256 * 4 + 512 * (4 + 2 * sizeof(R));
    if (bytes < etab) bytes = etab;
    return (bytes + 15) / 16 * 16;
}

template <typename R> struct RpSparseSlot {
    int* ctl;                    // [4]
    unsigned long long* rkey;    // [rec]
    R* rx; R* rd; R* out;        // [rec]
    int* perm; unsigned* okey;   // [rec]
    R* val; int* key;            // [nz]
    // energy table (aliases the record arrays): 256 partial sums x 2 entries
    int* members; int* ekey; R* ebefore; R* eafter;
    // per-row results (alias the record arrays once the chains have run)
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
    // energy table over the same bytes
    char* q = base + 16;
    L.ebefore = reinterpret_cast<R*>(q); q += 512 * sizeof(R);
    L.eafter = reinterpret_cast<R*>(q); q += 512 * sizeof(R);
    L.ekey = reinterpret_cast<int*>(q); q += 512 * 4;
    L.members = reinterpret_cast<int*>(q);
    // per-row cells over the factor / key arrays (rows <= rec)
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
        return staged_dict_bytes(P, A.sp) + (size_t)A.caps.teams * rp_sparse_slot_bytes<R>(A.caps);
    }
    static size_t total_lds_bytes(const DevParams& P, const Args& A) { return ((sizeof(Shared) + 15) / 16) * 16 + policy_lds_bytes(P, A); }
    static __device__ __forceinline__ char* slot_base(const DevParams& P, const Args& A, char* lds, int wv)
    {
        return lds + staged_dict_bytes(P, A.sp) + (size_t)wv * rp_sparse_slot_bytes<R>(A.caps);
    }
    static __device__ __forceinline__ const R* weights(const DevParams& P, const State<R>&, const Args& A, char* lds) { return dict_view(P, A.sp, lds).wts; }
    static __device__ __forceinline__ int units_per_atom(const DevParams&) { return 1; }
    static __device__ __forceinline__ int teams(const Args& A) { return A.caps.teams; }
    static __device__ __forceinline__ void after_atom(const DevParams&, const Args&, char*, int) {}
    static __device__ __forceinline__ void epilogue(const DevParams&, const State<R>&, const Args&, char*, int) {}

    static __device__ __forceinline__ void prologue(const DevParams& P, const State<R>&, const Sig<R>&, const Args& A, char* lds, int)
    {
        stage_dict(P, A.sp, lds);          // (strides of 256 threads: the threads beyond write the same values again)
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

    // ---- local energies of the atom (k, c) at t (:1002-1005): the listed cells of its clipped span, the atom's non-zeros
    //      applied to the copy; the pinned order is 256 strided partial sums, each sequential in the cell index.  With at most
    //      two cells per partial sum the order is free (0 + a = a, a + b = b + a); otherwise, or when a row list of the span
    //      has overflowed, the dense walk (the generic kernel's own loop) runs.  One wave; result in every lane.
    static __device__ __forceinline__ void span_energies(const DevParams& P, const State<R>& S, const Sig<R>& G, const SparseArgs<R>& A,
                                                         const RpSparseSlot<R>& L, int t, int k, R c, int lane, R& eb_out, R& ea_out)
    {
        const int T = P.T, F = P.F, W = P.W;
        int s, e, es;
        const int len = centered_span(T, W, t, s, e, es);
        const int* cnt = A.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A.rl_f + (int64_t)blockIdx.x * T * 8;
        const R nc = -c;
        fence();                                              // (the slot's previous user -- this wave -- is done with it)
#pragma unroll
        for (int u = 0; u < 4; ++u) L.members[lane + 64 * u] = 0;
        fence();
        bool bad = false;
        // the span's listed non-zero cells: a row's count and list in one round trip, its cells in the next
        for (int g0 = s; g0 < e; g0 += 64) {
            const int g = g0 + lane;
            const bool row_on = g < e;
            const int gq = row_on ? g : s;
            const int n = list_count(cnt + gq);
            const int4* row = reinterpret_cast<const int4*>(lf + (int64_t)gq * 8);
            const int4 a = row[0], b = row[1];
            if (row_on && n > 8) bad = true;
            const int fs[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const bool cells_on = row_on && n > 0 && n <= 8;
            R vs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vs[u] = (cells_on && fs[u] >= 0) ? G.r[(int64_t)g * F + fs[u]] : (R)0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (!(vs[u] != (R)0)) continue;
                const int i = (g - s) * F + fs[u];
                const int q = i & 255;
                const int at = atomicAdd(&L.members[q], 1);
                if (at < 2) { L.ekey[2 * q + at] = i; L.ebefore[2 * q + at] = vs[u]; L.eafter[2 * q + at] = vs[u]; }
            }
        }
        fence();
        // the atom's non-zeros: -c*D[k] rounded, then += (utils.py:120,129); a cell that is not listed yet starts from 0
        const int e0 = A.nzptr[k], e1 = A.nzptr[k + 1];
        for (int q0 = e0; q0 < e1; q0 += 64) {
            const int qe = q0 + lane;
            if (qe < e1) {
                const int wf = A.nzwf[qe], f = wf & 0xffff, g = t - P.off + (wf >> 16);
                if (g >= s && g < e) {                        // (the clipped part of the atom touches nothing)
                    const R prod = nc * A.nzval[qe];
                    const int i = (g - s) * F + f;
                    const int q = i & 255;
                    const int m = min(L.members[q], 2);
                    int hit = -1;
                    for (int j = 0; j < m; ++j) if (L.ekey[2 * q + j] == i) hit = j;
                    if (hit >= 0) {
                        L.eafter[2 * q + hit] = L.ebefore[2 * q + hit] + prod;
                    } else {
                        const int at = atomicAdd(&L.members[q], 1);
                        if (at < 2) { L.ekey[2 * q + at] = i; L.ebefore[2 * q + at] = (R)0; L.eafter[2 * q + at] = (R)0 + prod; }
                    }
                }
            }
        }
        fence();
        R pb[4], pa[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = lane + 64 * u;
            const int m = L.members[q];
            if (m > 2) bad = true;
            const R b0 = m > 0 ? L.ebefore[2 * q] : (R)0, b1 = m > 1 ? L.ebefore[2 * q + 1] : (R)0;
            const R a0 = m > 0 ? L.eafter[2 * q] : (R)0, a1 = m > 1 ? L.eafter[2 * q + 1] : (R)0;
            const R sb0 = b0 * b0, sb1 = b1 * b1, sa0 = a0 * a0, sa1 = a1 * a1;
            const R hb = (R)0 + sb0, ha = (R)0 + sa0;
            pb[u] = hb + sb1; pa[u] = ha + sa1;
        }
        if (__ballot(bad) != 0ull) {
            // dense walk: partial sum q = cell index mod 256, ascending (modeling.py:996-1016 as the generic kernel runs it)
            const int n = len * F;
            const R* dk = S.D + ((int64_t)k * W + es) * F;
            const R* rv = G.r + (int64_t)s * F;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                R b = (R)0, a = (R)0;
                for (int i = lane + 64 * u; i < n; i += 256) {
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
        for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const R ob = __shfl_down(pb[u], m), oa = __shfl_down(pa[u], m);
                pb[u] = pb[u] + ob; pa[u] = pa[u] + oa;
            }
        }
        const R b01 = pb[0] + pb[1], b23 = pb[2] + pb[3], a01 = pa[0] + pa[1], a23 = pa[2] + pa[3];
        eb_out = wave_bcast(b01 + b23, 0);
        ea_out = wave_bcast(a01 + a23, 0);
        fence();
    }

    static __device__ __forceinline__ void candidate(const DevParams& P, const State<R>& S, const Sig<R>& G, const Args& A0, char* lds,
                                                     int t, int lane, int wv, int& k_out, R& c_out, R& eb, R& ea, int& flag)
    {
        const SparseArgs<R> A = dict_view(P, A0.sp, lds);
        const RpSparseSlot<R> L = rp_sparse_slot<R>(slot_base(P, A0, lds, wv), A0.caps);
        k_out = __builtin_amdgcn_readfirstlane(G.bk[t]);
        c_out = wave_bcast(G.bc[t], 0);
        span_energies(P, S, G, A, L, t, k_out, c_out, lane, eb, ea);
        flag = is_interior(P, t) ? RPF_INTERIOR : 0;
    }

    static __device__ __forceinline__ void energies(const DevParams& P, const State<R>& S, const Sig<R>& G, const Args& A0, char* lds,
                                                    int t, int k, R c, int lane, int wv, R& eb, R& ea)
    {
        const SparseArgs<R> A = dict_view(P, A0.sp, lds);
        const RpSparseSlot<R> L = rp_sparse_slot<R>(slot_base(P, A0, lds, wv), A0.caps);
        span_energies(P, S, G, A, L, t, k, c, lane, eb, ea);
    }

    // ---- the atom's non-zeros into the residual (:996-1016 restricted to them: a zero of the atom changes nothing), its
    //      cells into the row lists (the window gathers read the lists instead of scanning rows of F values)
    static __device__ __forceinline__ void subtract(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A0, char* lds,
                                                    int p, int k, R c, int lane, int)
    {
        const SparseArgs<R> A = dict_view(P, A0.sp, lds);
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

    // ---- rows p-(W-1) .. p+(W-1) from the (final) residual, reflect padded (:1018-1051): sparse window x sparse dictionary
    static __device__ __forceinline__ void recorrelate(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A0, char* lds,
                                                       int p, int, int, bool interior, int lane, int wv)
    {
        const SparseArgs<R> A = dict_view(P, A0.sp, lds);
        const RpSparseSlot<R> L = rp_sparse_slot<R>(slot_base(P, A0, lds, wv), A0.caps);
        const int T = P.T, F = P.F, W = P.W, K = P.K;
        const int nzcap = A0.caps.nz, reccap = A0.caps.rec;
        const int nrows = 2 * W - 1, row0 = p - (W - 1), nwin = 3 * W - 2, g0 = row0 - P.off;
        const int tstart = p - P.off - (W - 1), tend = p + W / 2 + (W - 1);
        const int sidx = tstart < 0 ? 0 : tstart, eidx = tend > T - 1 ? T - 1 : tend, nslice = eidx - sidx + 1;
        const int* cnt = A.rl_cnt + (int64_t)blockIdx.x * T;
        const int* lf = A.rl_f + (int64_t)blockIdx.x * T * 8;
        fence();
        if (lane < 4) L.ctl[lane] = 0;
        fence();
        // 1. the non-zero cells of the window (any order): [0] count, [1] overflowed rows
        for (int j0 = 0; j0 < nwin; j0 += 64) {
            const int j = j0 + lane;
            if (j >= nwin) continue;
            const int gg = interior ? g0 + j : reflect_index(g0 + j, sidx, nslice);     // np.pad 'reflect', :1046
            const int n = list_count(cnt + gg);
            const int4* row = reinterpret_cast<const int4*>(lf + (int64_t)gg * 8);
            const int4 a = row[0], b = row[1];
            if (n > 8) { atomicAdd(&L.ctl[1], 1); continue; }
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
        fence();
        const int n = L.ctl[0];
        bool listed_ok = L.ctl[1] == 0 && n <= nzcap;
        int m = 0;
        if (listed_ok) {
            // 2. every input non-zero (f, j) with the dictionary non-zeros (k, w) of feature f: output row j - w
            int longest = 0;
            for (int i = lane; i < n; i += 64) {
                const int f = L.key[i] >> 16;
                longest = max(longest, A.fptr[f + 1] - A.fptr[f]);
            }
            longest = wave_max_i32(longest);
            const float inv = longest > 0 ? 1.0f / (float)longest : 0.0f;
            for (int it = lane; it < n * longest; it += 64) {
                int i = (int)(((float)it + 0.5f) * inv);                     // it / longest (exact: it < 2^17)
                int sl = it - i * longest;
                if (sl < 0) { --i; sl += longest; } else if (sl >= longest) { ++i; sl -= longest; }
                const int key = L.key[i], f = key >> 16, j = key & 0xffff;
                const int b = A.fptr[f], len = A.fptr[f + 1] - b;
                if (sl >= len) continue;
                const int kw = A.fkw[b + sl], w = kw & 0xffff, row = j - w;
                const R d = A.fval[b + sl];
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
            fence();
            m = L.ctl[2];
            if (m > reccap) listed_ok = false;
        }
        if (listed_ok) {
            // 3. sort by (output, chain order); the keys are distinct (rank sort, eight keys per LDS round trip)
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
            // 5. per-row best over atoms: the listed outputs against the zeros of all the others -- a zero score never
            //    beats k = 0, the first of the ties; among equal scores the lowest atom wins (np.argmax)
            for (int row = lane; row < nrows; row += 64) { L.rmax[row] = 0ull; L.rk[row] = INT_MAX; L.c0[row] = (R)0; }
            fence();
            for (int sp = lane; sp < m; sp += 64) {
                const unsigned ok = L.okey[sp];
                if (ok == ~0u) continue;
                const int row = (int)(ok >> 16), kk = (int)(ok & 0xffffu);
                const R o = L.out[sp];
                if (kk == 0) L.c0[row] = o;                                  // the value of the default winner
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
            for (int row = lane; row < nrows; row += 64) {
                const int t = row0 + row;
                if (t < 0 || t >= T) continue;                               // overlapReplace clipping (utils.py:133-161)
                const int k = L.rk[row];
                if (k == INT_MAX) { G.bc[t] = L.c0[row]; G.bk[t] = 0; }
                else { G.bc[t] = L.rc[row]; G.bk[t] = k; }
            }
            fence();
            return;
        }
        // The lists do not fit this wave's slot (or a row list of the window has overflowed): each output walks its atom's
        // non-zeros (already in chain order), row by row, the lanes over the atoms.
        for (int row = 0; row < nrows; ++row) {
            const int t = row0 + row;
            if (t < 0 || t >= T) continue;                                   // (uniform)
            Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
            R bc = (R)0;
            for (int k = lane; k < K; k += 64) {
                R acc = (R)0;
                const int e1 = A.nzptr[k + 1];
                for (int e = A.nzptr[k]; e < e1; ++e) {
                    const int wf = A.nzwf[e];
                    const int g = interior ? g0 + row + (wf >> 16) : reflect_index(g0 + row + (wf >> 16), sidx, nslice);
                    acc = rfma(G.r[(int64_t)g * F + (wf & 0xffff)], A.nzval[e], acc);
                }
                const R sc = score_of(acc, k, A.wts);
                if (sc > best.s) { best.s = sc; best.i = k; bc = acc; }       // ascending k per lane: '>' keeps the first of equals
            }
            const Cand<R> win = wave_argmax(best);
            const int owner = __ffsll((long long)__ballot(best.i == win.i)) - 1;
            const R wc = wave_bcast(bc, owner);
            if (lane == 0) { G.bc[t] = wc; G.bk[t] = win.i; }
        }
        fence();
    }
};

// host-side dispatch -----------------------------------------------------------------------------
// LDS slots of the waves: as many as fit beside the control block and the staged dictionary lists, at most one per block
// of the round (more waves than atoms have nothing to do)
template <typename R> inline RpSparseCaps rp_sparse_caps(const DevParams& P, const SparseArgs<R>& sp, size_t fixed_bytes)
{
    RpSparseCaps c;
    auto pow2 = [](int v) { int p = 1; while (p < v) p <<= 1; return p; };
    c.rows = 2 * P.W - 1;
    c.nz = std::min(256, std::max(64, pow2(3 * P.W - 2)));
    c.rec = std::max(pow2(c.rows), std::min(512, std::max(128, pow2(3 * (3 * P.W - 2)))));
    const size_t budget = (size_t)160 * 1024 - fixed_bytes - staged_dict_bytes(P, sp);
    for (;;) {
        c.teams = (int)std::min<size_t>(kRpWaves, budget / rp_sparse_slot_bytes<R>(c));
        if (c.teams >= std::min(kRpWaves, std::max(4, P.maxsel)) || c.rec <= pow2(c.rows) || c.rec <= 128) break;
        c.rec /= 2; c.nz = std::max(64, c.nz / 2);
    }
    return c;
}

template <typename R>
static int rp_sparse_launch(hipStream_t stream, const DevParams& P0, const State<R>& S, const SparseArgs<R>& sp, bool dry)
{
    using Pol = RpSparse<R>;
    if (!rp_params_ok(P0, Pol::kMaxSel)) return -1;
    // the row lists (8 features per row) and the per-atom / by-feature dictionary lists carry this policy
    if (!sp.rl_cnt || sp.rl_cap != 8 || !sp.nzptr || !sp.fptr || P0.W > 16384 || 3 * P0.W - 2 > 0xffff || P0.K > 65535) return -1;
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    RpSparseArgs<R> A;
    A.sp = sp;
    A.caps = rp_sparse_caps<R>(P, sp, ((sizeof(typename Pol::Shared) + 15) / 16) * 16);
    if (A.caps.teams < 2 || A.caps.rows > A.caps.rec) return -1;
    const size_t lds = Pol::total_lds_bytes(P, A);
    if (lds > (size_t)160 * 1024) return -1;
    auto kern = iterate_rp_kernel<R, Pol>;
    if (set_dyn_lds((const void*)kern, lds) != hipSuccess) return -1;
    if (dry) return 0;
    hipLaunchKernelGGL(kern, dim3(P.B), dim3(kRpThreads), lds, stream, P, S, A);
    return 0;
}

}  // namespace hscmp
