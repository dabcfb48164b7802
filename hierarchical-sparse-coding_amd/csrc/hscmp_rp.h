// hscmp_rp.h -- round-parallel form of the greedy loop, for BLOCKED selection (modeling.py:908-963).
//
// iterate_kernel (hscmp_kernels.h) gives a signal ONE team of 256 threads that applies the atoms of a selection
// round one after the other (modeling.py:1101-1120): at small batches -- BASELINE config 5 is 128 signals per GPU,
// ~17 k atoms per signal and level -- the run time is the latency of one signal's chain of atoms and most of the chip
// idles.  A blocked round, however, hands out up to nb+1 atoms at once, and once the interference filter (:951-957)
// has applied they are pairwise >= W apart: their supports are disjoint.  Then
//   * the local energies before / after an atom's subtraction (:1002-1005) do not depend on the other atoms of the
//     round, so the whole bookkeeping chain of the round -- duplicates / nnz (:1106-1111), residual energy (:1014), the
//     fast stop rules after every atom (:1125-1142), which may cut the round short -- is a PREFIX over per-atom values
//     that are all known before anything is written;
//   * a re-correlated row (:1018-1051) reads residual samples that only atoms within W-1 of it change, and each of
//     those atoms re-correlates the row itself: "all subtractions, then all re-correlations from the final residual"
//     leaves every row exactly as the sequential loop does (rows that two atoms share are written twice with the same
//     value).
// So here a signal owns a 1024-thread workgroup (16 waves, a CU to itself) and a round runs as phases over ALL its atoms:
//   P1  one wave per block: arg-max of the block (:935-937), its (k, c), local energies, coefficient-slot lookup
//   P2  null / interference filters, |c| order, weak-atom filter (:946-962, :1090-1099) on index lists in LDS
//   P3  prefix by one thread: bookkeeping and stop rules in selection order -> how many atoms of the round apply
//   P4  one wave per atom: residual subtraction                                  | workgroup barrier
//   P5  one wave per (atom, row tile): re-correlation from the final residual    | workgroup barrier
//   P6  maxima of the touched segments, slow stop rules (:1145-1163)
// What is order dependent stays sequential, and exact by construction: an atom whose window crosses a signal end
// (reflect padding, the stale-sample quirk of DESIGN.md section 2) forms a group of its own inside the round, and a
// round whose interference filter was skipped (no gap qualifies: its atoms may overlap) runs atom by atom with the
// energies taken when each atom's turn comes.  Same results as iterate_kernel, bit for bit
// (tests/test_gpu_round_parallel.py runs both on every case).
#pragma once

#include "hscmp_mfma.h"

namespace hscmp {

constexpr int kRpWaves = 16;
constexpr int kRpThreads = 64 * kRpWaves;

enum { RPF_INTERIOR = 1 };      // candidate flags: the atom's 3W-2 window lies inside the signal and no edge quirk applies

template <typename R, int MAXSEG, int MAXSEL> struct RpShared {
    R seg_score[MAXSEG];
    int seg_t[MAXSEG];
    // candidates of the round, one per block (:935-937)
    int c_t[MAXSEL]; int c_k[MAXSEL]; R c_c[MAXSEL];
    R c_eb[MAXSEL]; R c_ea[MAXSEL];           // energy of the atom's clipped window before / after its subtraction
    int c_found[MAXSEL]; double c_acc[MAXSEL];  // coefficient slot of (t, k) as of the round start (-1: none) and its accumulator
    int c_flag[MAXSEL];
    int c_head[MAXSEL];                         // first link of the position's slot chain as of the round start
    int idx[2][MAXSEL];                         // index lists of the filters (ping-pong)
    int ord[MAXSEL];                            // the round's atoms in application order

    int wtot[kRpWaves];
    // control block (written by thread 0, read by all behind a barrier)
    int converged, stop, napply, gend;
    int n, spaced, nedge, full;   // the round's atom count; pairwise >= W apart; atoms at a signal end; event list too short
    int nnz, ndup, rounds, iters, nev, nslots, offset;
    R e_sig, e_res;
};

// stable compaction over the workgroup: thread i keeps `value` iff `keep`; dst[0..count) in thread order; returns count
template <typename SH> __device__ __forceinline__ int rp_compact(SH& sh, bool keep, int value, int* dst)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) sh.wtot[wv] = __popcll(mask);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < kRpWaves; ++q) { const int c = sh.wtot[q]; if (q < wv) before += c; total += c; }
    if (keep) dst[before + __popcll(mask & ((1ull << lane) - 1ull))] = value;
    __syncthreads();
    return total;
}

// ---- wave reductions on the vector ALU for float64 (the generic forms of hscmp_device.h go through ds_bpermute: two to
//      three LDS-crossbar operations per step and value) ------------------------------------------------------------
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)          // result in every lane (wave-uniform)
{
    asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// (score, index) arg-max as wave_argmax: larger score, then the smaller index.  Scores are >= 0 (or -1 for "nothing"): their
// bit patterns order like the values (the sentinel is negative as an integer), so the maximum is taken word by word.
template <typename R> __device__ __forceinline__ Cand<R> rp_wave_argmax(Cand<R> c) { return wave_argmax(c); }
template <> __device__ __forceinline__ Cand<double> rp_wave_argmax<double>(Cand<double> c)
{
    const long long bits = __double_as_longlong(c.s);
    const int hi = (int)(bits >> 32);
    const unsigned lo = (unsigned)bits;
    const int mh = wave_max_i32(hi);
    const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
    Cand<double> r;
    r.s = __longlong_as_double(((long long)mh << 32) | (long long)ml);
    r.i = wave_min_i32((hi == mh && lo == ml) ? c.i : INT_MAX);
    return r;
}
// halving tree of the pinned sums (wave_tree_down2 of hscmp_device.h: lane i += lane i+m for m = 32 .. 1, total in lane 0)
// for float64: the partner's two words come through v_permlane32_swap / v_permlane16_swap and row_shl DPP moves
__device__ __forceinline__ void rp_tree_down2(float& a, float& b) { wave_tree_down2(a, b); }
__device__ __forceinline__ void rp_tree_down2(double& a, double& b)
{
    auto partner32 = [](double v) {
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
        return __hiloint2double((int)hi[1], (int)lo[1]);
    };
    auto partner16 = [](double v) {
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
        return __hiloint2double((int)hi[1], (int)lo[1]);
    };
    { const double oa = partner32(a), ob = partner32(b); a = a + oa; b = b + ob; }
    { const double oa = partner16(a), ob = partner16(b); a = a + oa; b = b + ob; }
#define HSCMP_RP_TREE_STEP(CTRL)                                                                                               \
    {                                                                                                                          \
        const double oa = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(a), CTRL, 0xF, 0xF, true),            \
                                           __builtin_amdgcn_update_dpp(0, __double2loint(a), CTRL, 0xF, 0xF, true));           \
        const double ob = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(b), CTRL, 0xF, 0xF, true),            \
                                           __builtin_amdgcn_update_dpp(0, __double2loint(b), CTRL, 0xF, 0xF, true));           \
        a = a + oa; b = b + ob;                                                                                                \
    }
    HSCMP_RP_TREE_STEP(0x108)      // row_shl:8
    HSCMP_RP_TREE_STEP(0x104)
    HSCMP_RP_TREE_STEP(0x102)
    HSCMP_RP_TREE_STEP(0x101)
#undef HSCMP_RP_TREE_STEP
}

// arg-max of the per-position best over [t0, t1) by one wave, t1 - t0 <= 256: the (up to four) loads of a lane are issued
// together -- one memory round trip, where a load / compare loop makes one per 64 positions.  Same result as
// wave_range_argmax (maximum score, lowest position among equals).
template <bool SO, typename R>
__device__ __forceinline__ Cand<R> rp_range_argmax4(const Sig<R>& G, const R* w, int t0, int t1, int lane)
{
    R sc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = t0 + lane + 64 * u;
        sc[u] = (R)-1;
        if (t < t1) { if constexpr (SO) sc[u] = G.bc[t]; else sc[u] = score_of(G.bc[t], G.bk[t], w); }
    }
    Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
#pragma unroll
    for (int u = 0; u < 4; ++u) {                       // ascending position per lane: '>' keeps the first of equals
        const int t = t0 + lane + 64 * u;
        if (t < t1 && sc[u] > best.s) { best.s = sc[u]; best.i = t; }
    }
    return rp_wave_argmax(best);
}

// segment maximum of the per-position best (score, first position) by one wave
template <bool SO, typename R, typename SH>
__device__ __forceinline__ void rp_scan_segment(const DevParams& P, const Sig<R>& G, const R* w, SH& sh, int sg, int lane)
{
    const int t0 = (sg << P.seg_shift);
    const int t1 = min(P.T, t0 + P.seg);
    Cand<R> win = P.seg <= 256 ? rp_range_argmax4<SO>(G, w, t0, t1, lane) : wave_range_argmax<SO>(G, w, t0, t1, lane);
    if (lane == 0) {
        if (win.i == INT_MAX) { win.i = t0; win.s = (R)0; }
        sh.seg_score[sg] = win.s;
        sh.seg_t[sg] = win.i;
    }
}

// two neighbouring segments of at most 128 positions each by one wave: their loads in ONE batch, then the two arg-maxes
template <bool SO, typename R, typename SH>
__device__ __forceinline__ void rp_scan_segment_pair(const DevParams& P, const Sig<R>& G, const R* w, SH& sh, int sg, int lane)
{
    const int t0 = (sg << P.seg_shift), tm = t0 + P.seg, t1 = min(P.T, tm + P.seg);
    R sc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = (u < 2 ? t0 : tm) + lane + 64 * (u & 1);
        const int tend = u < 2 ? tm : t1;
        sc[u] = (R)-1;
        if (t < tend && (t - (u < 2 ? t0 : tm)) < P.seg) { if constexpr (SO) sc[u] = G.bc[t]; else sc[u] = score_of(G.bc[t], G.bk[t], w); }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int base = h ? tm : t0, tend = h ? t1 : tm;
        Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
#pragma unroll
        for (int u = 0; u < 2; ++u) {                    // ascending position per lane: '>' keeps the first of equals
            const int t = base + lane + 64 * u;
            if (t < tend && lane + 64 * u < P.seg && sc[2 * h + u] > best.s) { best.s = sc[2 * h + u]; best.i = t; }
        }
        Cand<R> win = rp_wave_argmax(best);
        if (lane == 0) {
            if (win.i == INT_MAX) { win.i = base; win.s = (R)0; }
            sh.seg_score[sg + h] = win.s;
            sh.seg_t[sg + h] = win.i;
        }
    }
}

// arg-max of the per-position best over the block [lo, hi) by one wave through the segment maxima (wave_block_argmax of
// hscmp_kernels.h), the two ragged ends -- less than a segment each -- fetched in ONE batch of loads.
template <bool SO, typename R, typename SH>
__device__ __forceinline__ Cand<R> rp_block_argmax(const DevParams& P, const Sig<R>& G, const R* w, const SH& sh, int lo, int hi, int lane)
{
    const int sgA = (lo + P.seg - 1) >> P.seg_shift;                        // first segment entirely inside
    const int sgB = (hi >= P.T) ? P.nseg - 1 : (hi >> P.seg_shift) - 1;     // last one (the last segment of the signal may be short)
    if (P.seg > 128 || sgA > sgB) return wave_block_argmax<SO>(P, G, w, sh, lo, hi, lane);
    const int headEnd = sgA << P.seg_shift, tailBegin = min(hi, (sgB + 1) << P.seg_shift);
    R hs[2], ts[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int th = lo + lane + 64 * u, tt = tailBegin + lane + 64 * u;
        hs[u] = (R)-1; ts[u] = (R)-1;
        if (th < headEnd) { if constexpr (SO) hs[u] = G.bc[th]; else hs[u] = score_of(G.bc[th], G.bk[th], w); }
        if (tt < hi) { if constexpr (SO) ts[u] = G.bc[tt]; else ts[u] = score_of(G.bc[tt], G.bk[tt], w); }
    }
    Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
    // per lane the candidates come in ascending position, so '>' keeps the first of equals
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int th = lo + lane + 64 * u; if (th < headEnd && hs[u] > best.s) { best.s = hs[u]; best.i = th; } }
    for (int sg = sgA + lane; sg <= sgB; sg += 64) {
        const R sc = sh.seg_score[sg];
        if (sc > best.s) { best.s = sc; best.i = sh.seg_t[sg]; }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int tt = tailBegin + lane + 64 * u; if (tt < hi && ts[u] > best.s) { best.s = ts[u]; best.i = tt; } }
    return rp_wave_argmax(best);
}

// ---- P2 by the whole workgroup (more than 64 blocks: nbBlocks='auto' on long signals): index lists in LDS --------
template <typename R, typename SH>
__device__ __forceinline__ void rp_block_select(const DevParams& P, SH& sh, int nb)
{
    const int tid = threadIdx.x, T = P.T, W = P.W, F = P.F;
    const bool has_thres = P.has_thres != 0;
    const double thres = P.thres;
    constexpr int MAXSEL = (int)(sizeof(sh.ord) / sizeof(int));
    int* la = sh.idx[0]; int* lb = sh.idx[1];
    bool spaced = true;
    // :946-948 drop null coefficients (and invalid blocks)
    int n = rp_compact(sh, tid < nb && sh.c_t[tid] >= 0 && (!has_thres || fabs((double)sh.c_c[tid]) > thres), tid, la);
    // :951-957 interference filter vs the unfiltered predecessor; skipped when no gap qualifies
    if (n > 1) {
        const bool gap = tid >= 1 && tid < n && (sh.c_t[la[tid]] - sh.c_t[la[tid - 1]] >= W);
        if (__syncthreads_or(gap ? 1 : 0)) {
            n = rp_compact(sh, tid < n && (tid == 0 || gap), tid < n ? la[tid] : 0, lb);
            int* tmp = la; la = lb; lb = tmp;
        } else {
            spaced = false;
        }
    }
    // :960-962 argsort(|c|)[::-1]: descending, the later entry first among equals (rank sort)
    if (tid < n) {
        const int me = la[tid];
        const R a = rabs(sh.c_c[me]);
        int rank = 0;
        for (int q = 0; q < n; ++q) {
            const R o = rabs(sh.c_c[la[q]]);
            rank += (o > a || (o == a && q > tid)) ? 1 : 0;
        }
        lb[rank] = me;
    }
    __syncthreads();
    { int* tmp = la; la = lb; lb = tmp; }
    // :1090-1099 weak-atom filter: the window of the filter is the atom's clipped support, its energy c_eb
    if (P.has_snr && n > 1) {
        const R tol_energy = sh.e_sig / (R)P.snr_ratio;
        const double thr = (double)tol_energy / (double)((int64_t)T * F);
        bool keep = false;
        if (tid < n) {
            const int me = la[tid];
            int s, e, es;
            const int len = centered_span(T, W, sh.c_t[me], s, e, es);
            const R mean = sh.c_eb[me] / (R)((int64_t)len * F);
            keep = (double)mean >= thr;
        }
        n = rp_compact(sh, keep, tid < MAXSEL ? la[tid] : 0, lb);
        int* tmp = la; la = lb; lb = tmp;
    }
    if (tid < n) sh.ord[tid] = la[tid];
    const int nedge = __syncthreads_count((tid < n && !(sh.c_flag[sh.ord[tid]] & RPF_INTERIOR)) ? 1 : 0);
    if (tid == 0) { sh.n = n; sh.spaced = spaced ? 1 : 0; sh.nedge = nedge; }
    __syncthreads();
}

// ---- P2 by ONE wave (at most 64 blocks): candidate j stays in lane j; the filters are ballot masks, the |c| order a rank
//      among the surviving lanes over v_readlane broadcasts (the weak-atom filter keeps the order of the sort it follows,
//      so the final rank is the rank among its survivors).  One batch of LDS reads, one lane exchange, one write. --------
template <typename R, typename SH>
__device__ __forceinline__ void rp_wave_select(const DevParams& P, SH& sh, int nb, int lane)
{
    const int T = P.T, W = P.W, F = P.F;
    const bool mine = lane < nb;
    const int t = mine ? sh.c_t[lane] : -1;
    const R c = mine ? sh.c_c[lane] : (R)0;
    const R eb = mine ? sh.c_eb[lane] : (R)0;
    const int flag = mine ? sh.c_flag[lane] : RPF_INTERIOR;
    const R e_sig = sh.e_sig;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // :946-948 drop null coefficients (and invalid blocks)
    unsigned long long mask = __ballot(t >= 0 && (P.has_thres == 0 || fabs((double)c) > P.thres));
    bool spaced = true;
    // :951-957 interference filter vs the unfiltered predecessor; skipped when no gap qualifies
    if (__popcll(mask) > 1) {
        const unsigned long long before = mask & lt;
        const int pl = before ? 63 - __clzll((long long)before) : lane;
        const int tp = __shfl(t, pl);
        const unsigned long long gaps = __ballot(((mask >> lane) & 1ull) && before != 0ull && (t - tp >= W));
        if (gaps != 0ull) mask = gaps | (mask & (0ull - mask));                // (the first candidate always stays)
        else spaced = false;
    }
    // :1090-1099 weak-atom filter: the window of the filter is the atom's clipped support, its energy c_eb
    if (P.has_snr && __popcll(mask) > 1) {
        const R tol_energy = e_sig / (R)P.snr_ratio;
        const double thr = (double)tol_energy / (double)((int64_t)T * F);
        int s, e, es;
        const int len = centered_span(T, W, t < 0 ? 0 : t, s, e, es);
        const R mean = eb / (R)((int64_t)len * F);
        mask = __ballot(((mask >> lane) & 1ull) && (double)mean >= thr);
    }
    // :960-962 argsort(|c|)[::-1]: descending, the later entry first among equals
    const bool in = ((mask >> lane) & 1ull) != 0ull;
    const R a = rabs(c);
    int rank = 0;
    for (unsigned long long rest = mask; rest; rest &= rest - 1ull) {        // (uniform)
        const int q = __ffsll((long long)rest) - 1;
        const R o = wave_bcast(a, q);
        rank += (o > a || (o == a && q > lane)) ? 1 : 0;
    }
    if (in) sh.ord[rank] = lane;
    const unsigned long long edges = __ballot(in && !(flag & RPF_INTERIOR));
    if (lane == 0) { sh.n = __popcll(mask); sh.spaced = spaced ? 1 : 0; sh.nedge = __popcll(edges); }
}

// ---- P3 by ONE wave: bookkeeping (:1106-1114) and the fast stop rules (:1125-1142) of the atoms ord[pos, gend) in
//      selection order, 64 atoms at a time, atom i in lane i.  The counters are prefix popcounts; the residual energy is
//      the same chain of subtractions (:1014), carried through v_readlane; every lane tests the stop rules of its own atom
//      and the first lane that stops cuts the group.  The lanes of the applied atoms write their slots and events. ------
struct RpPending { int t, k, si, ev, fresh, on, head; double a; };
template <typename R> __device__ __forceinline__ void rp_store_pending(const DevParams& P, const Sig<R>& G, const RpPending& q, R c)
{
    if (!q.on) return;
    if (q.fresh) { G.slot_t[q.si] = q.t; G.slot_k[q.si] = q.k; hval_store(G.hval + q.si, q.head); hval_store(G.head + q.t, q.si); }
    G.slot_a[q.si] = q.a;
    G.ev_t[q.ev] = q.t; G.ev_k[q.ev] = q.k; G.ev_c[q.ev] = c;
}
// (the slot / event stores of the LAST chunk are handed back: the caller issues them behind the workgroup barrier, where
//  the other waves are already at work)
template <typename R, typename SH>
__device__ __forceinline__ void rp_wave_prefix(const DevParams& P, SH& sh, const Sig<R>& G, int pos, int gend, int lane, RpPending& pend, R& pend_c)
{
    pend.on = 0; pend_c = (R)0;
    int nnz = sh.nnz, ndup = sh.ndup, nslots = sh.nslots, nev = sh.nev, iters = sh.iters;
    R e_res = sh.e_res;
    const R e_sig = sh.e_sig;
    int conv = 0, stop = STOP_RUNNING, applied_end = gend;
    const unsigned long long lt = (1ull << lane) - 1ull, le = lt | (1ull << lane);
    for (int base = pos; base < gend && !conv; base += 64) {
        const int cnt = min(64, gend - base);
        const bool on = lane < cnt;
        const int me = on ? sh.ord[base + lane] : 0;
        const R c = sh.c_c[me];
        const int found = sh.c_found[me];
        const double acc = sh.c_acc[me];
        const R loss = on ? sh.c_eb[me] - sh.c_ea[me] : (R)0;        // :1005
        const bool is_dup = on && found >= 0 && fabs(acc) > 0.0;
        const bool is_nnz = on && !is_dup && rabs(c) > (R)0;
        const bool fresh = on && found < 0;
        const unsigned long long m_nnz = __ballot(is_nnz), m_dup = __ballot(is_dup), m_fresh = __ballot(fresh);
        R cur = e_res, mine = (R)0;
        for (int j = 0; j < cnt; ++j) {                              // :1014, atom after atom
            cur = cur - wave_bcast(loss, j);
            if (lane == j) mine = cur;
        }
        const int nnz_i = nnz + __popcll(m_nnz & le);
        int st = STOP_RUNNING;
        if (on) {
            if ((double)mine < P.eps) st = STOP_ENERGY_EPS;
            else if (P.l0 >= 0 && nnz_i >= P.l0) st = STOP_NNZ;
            else if (P.has_snr) {
                const R q = e_sig / mine;
                if ((double)q >= P.snr_ratio) st = STOP_SNR;
            }
        }
        const unsigned long long ms = __ballot(st != STOP_RUNNING);
        const int take = ms ? __ffsll((long long)ms) : cnt;          // atoms of this chunk that are applied
        const unsigned long long below = take >= 64 ? ~0ull : ((1ull << take) - 1ull);
        // coefficient slot and event of the atom (:1114; :992: a new accumulator starts at 0.0)
        pend.on = lane < take; pend.t = sh.c_t[me]; pend.k = sh.c_k[me]; pend.fresh = fresh; pend.head = sh.c_head[me];
        pend.si = fresh ? nslots + __popcll(m_fresh & lt) : found; pend.ev = nev + lane; pend.a = acc + (double)c; pend_c = c;
        if (!ms && base + 64 < gend) { rp_store_pending<R>(P, G, pend, pend_c); pend.on = 0; }
        nnz += __popcll(m_nnz & below); ndup += __popcll(m_dup & below); nslots += __popcll(m_fresh & below);
        nev += take; iters += take;
        e_res = wave_bcast(mine, take - 1);
        if (ms) { conv = 1; stop = __builtin_amdgcn_readlane(st, take - 1); applied_end = base + take; }
    }
    if (lane == 0) {
        sh.nnz = nnz; sh.ndup = ndup; sh.nslots = nslots; sh.nev = nev; sh.iters = iters; sh.e_res = e_res;
        if (conv) { sh.converged = 1; sh.stop = stop; }
        sh.napply = applied_end; sh.gend = gend;
    }
}

// ------------------------------------------------------------------------------------------------
// the round-parallel loop          grid = B, block = kRpThreads
//   Pol supplies the per-wave pieces: candidate (resolve + energies), energies, subtract, recorrelate.
// ------------------------------------------------------------------------------------------------
template <typename R, typename Pol>
__global__ __launch_bounds__(kRpThreads) void iterate_rp_kernel(DevParams P, State<R> S, typename Pol::Args A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using SH = typename Pol::Shared;
    SH& sh = *reinterpret_cast<SH*>(smem);
    char* plds = smem + ((sizeof(SH) + 15) / 16) * 16;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* stats = S.stats + (int64_t)b * ST_COUNT;
    if (stats[ST_STOP] != STOP_RUNNING) return;          // converged in an earlier launch (uniform)

    const int T = P.T, W = P.W, F = P.F;
    Sig<R> G;
    G.r = S.residual + (int64_t)b * T * F;
    G.bc = S.best_c + (int64_t)b * T;
    G.bk = S.best_k + (int64_t)b * T;
    G.ev_t = S.ev_t + (int64_t)b * P.cap; G.ev_k = S.ev_k + (int64_t)b * P.cap; G.ev_c = S.ev_c + (int64_t)b * P.cap;
    G.slot_t = S.slot_t + (int64_t)b * P.cap; G.slot_k = S.slot_k + (int64_t)b * P.cap; G.slot_a = S.slot_a + (int64_t)b * P.cap;
    G.hkey = S.hkey + (int64_t)b * ((int64_t)P.hmask + 1); G.hval = S.hval + (int64_t)b * ((int64_t)P.hmask + 1);
    G.head = S.head + (int64_t)b * T;
    G.sel_t = nullptr; G.sel_k = nullptr; G.sel_c = nullptr;

    Pol::prologue(P, S, G, A, plds, b);                  // (ends with a workgroup barrier)
    const int nteams = Pol::teams(A);                    // waves that take per-atom work (each needs a strip of LDS)
    const R* wts = Pol::weights(P, S, A, plds);

    // ---- segment maxima; the (t,k) -> coefficient slot lookup of this launch (:1106-1114 looks the pair up in the dict of
    //      coefficients): the slots of a position are chained from head[t] (most recent first) through next[] = hval[] -- the
    //      address of the first link depends on the position only, so it is fetched beside the position's window, and a
    //      position that was never selected costs no second round trip.  A round selects every position at most once, so
    //      its new links are plain stores.
    for (int sg = wv; sg < P.nseg; sg += kRpWaves) rp_scan_segment<Pol::kScoreOnly>(P, G, wts, sh, sg, lane);
    for (int i = tid; i < T; i += kRpThreads) hval_store(G.head + i, -1);
    if (tid == 0) {
        sh.nnz = stats[ST_NNZ]; sh.ndup = stats[ST_DUP]; sh.rounds = stats[ST_ROUNDS]; sh.iters = stats[ST_ITERS];
        sh.nev = stats[ST_EVENTS]; sh.nslots = stats[ST_SLOTS]; sh.offset = stats[ST_OFFSET];
        sh.converged = 0; sh.stop = STOP_RUNNING; sh.napply = 0; sh.gend = 0; sh.n = 0; sh.spaced = 1; sh.nedge = 0; sh.full = 0;
        sh.e_sig = S.energy[2 * b + 0]; sh.e_res = S.energy[2 * b + 1];
    }
    __syncthreads();
    {
        const int ns = stats[ST_SLOTS];
        for (int i = tid; i < ns; i += kRpThreads)
            hval_store(G.hval + i, __hip_atomic_exchange(G.head + G.slot_t[i], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();

    constexpr int SB = Pol::kScoreOnly ? 16 : 0;      // (diagnostic build: stamp slots of the two policies)
    (void)SB;
    HSCMP_STAMP_BEGIN();
    for (int round = 0; P.max_rounds <= 0 || round < P.max_rounds; ++round) {
        HSCMP_STAMP(SB + 0);                                                      // round end of the previous round
        // =========================== P1: one arg-max per block (:908-937) ===========================
        const int off = sh.offset;
        const int nb = P.nbk + (off ? 1 : 0);
        const int pad0 = off ? P.bs / 2 : 0;
        for (int j = wv; j < nb && wv < nteams; j += nteams) {
            const int w0 = j * P.bs - pad0;
            const int lo = w0 < 0 ? 0 : w0;
            const int hi = min(T, w0 + P.bs);
            Cand<R> win; win.s = (R)-1; win.i = INT_MAX;
            if (lo < hi) win = rp_block_argmax<Pol::kScoreOnly>(P, G, wts, sh, lo, hi, lane);
            HSCMP_STAMP(SB + 9);
            bool valid = (lo < hi) && win.i != INT_MAX;                      // :940-942 range test
            if (valid && win.s == (R)0 && w0 < 0) valid = false;             // arg-max on a leading padded row
            int wk = 0, flag = 0, found = -1;
            R wc = (R)0, eb = (R)0, ea = (R)0;
            double acc = 0.0;
            int head = -1;
            if (valid) {                                                     // wave-uniform
                head = hval_load(G.head + win.i);                            // (in flight under the candidate's own loads)
                Pol::candidate(P, S, G, A, plds, win.i, lane, wv, wk, wc, eb, ea, flag);
                HSCMP_STAMP(SB + 10);
                found = head;
                while (found >= 0 && G.slot_k[found] != wk) found = hval_load(G.hval + found);      // (uniform)
                if (found >= 0) acc = G.slot_a[found];
                HSCMP_STAMP(SB + 11);
            }
            if (lane == 0) {
                sh.c_t[j] = valid ? win.i : -1; sh.c_k[j] = wk; sh.c_c[j] = wc; sh.c_eb[j] = eb; sh.c_ea[j] = ea;
                sh.c_found[j] = found; sh.c_acc[j] = acc; sh.c_flag[j] = flag; sh.c_head[j] = head;
            }
        }
        __syncthreads();
        HSCMP_STAMP(SB + 1);                                                      // block arg-max + candidates
        // =========================== P2: filters and order (:946-962, :1090-1099) ===========================
        // Up to 64 blocks: wave 0 alone, candidates in its lanes (no workgroup barrier until the atoms are known AND the
        // bookkeeping prefix of the first group is done); more blocks: the whole workgroup over index lists.
        const bool small = nb <= 64;
        if (!small) rp_block_select<R>(P, sh, nb);                           // (ends with a barrier)
        // =========================== P3-P5: apply the atoms (:1101-1142), group by group ===========================
        int pos = 0, n = 0;
        bool full = false;
        for (;;) {
            RpPending pend; pend.on = 0;
            R pend_c = (R)0;
            if (wv == 0) {
                if (pos == 0 && small) rp_wave_select<R>(P, sh, nb, lane);
                HSCMP_STAMP(SB + 7);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int nn = sh.n;
                // A round whose atoms do not all fit the event list is not started: the state then is exactly that of a
                // round boundary, and hscmp_grow_events + hscmp_continue resume bit for bit.
                if (pos == 0 && sh.nev + nn > P.cap) { if (lane == 0) sh.full = 1; }
                else if (pos < nn) {
                    int gend;
                    if (!sh.spaced) {
                        // overlapping atoms: one at a time, energies as of its turn
                        gend = pos + 1;
                        if (pos > 0) {
                            const int me = sh.ord[pos];
                            R eb, ea;
                            Pol::energies(P, S, G, A, plds, sh.c_t[me], sh.c_k[me], sh.c_c[me], lane, wv, eb, ea);
                            if (lane == 0) { sh.c_eb[me] = eb; sh.c_ea[me] = ea; }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_wave_barrier();
                        }
                    } else if (sh.nedge == 0) {
                        gend = nn;
                    } else if (!(sh.c_flag[sh.ord[pos]] & RPF_INTERIOR)) {
                        gend = pos + 1;                                      // an atom at a signal end: alone
                    } else {
                        gend = pos + 1;
                        while (gend < nn && (sh.c_flag[sh.ord[gend]] & RPF_INTERIOR)) ++gend;
                    }
                    rp_wave_prefix<R>(P, sh, G, pos, gend, lane, pend, pend_c);   // (-> sh.napply, sh.gend, sh.converged / sh.stop)
                    HSCMP_STAMP(SB + 8);
                }
            }
            __syncthreads();
            if (wv == 0) rp_store_pending<R>(P, G, pend, pend_c);
            n = sh.n;
            if (sh.full) { full = true; break; }
            if (pos >= n) break;
            const int aend = sh.napply, gend = sh.gend;
            HSCMP_STAMP(SB + 2);                                                  // filters + order + prefix
            // ---- P4: residual subtraction (:1117, :996-1016), one wave per atom
            for (int i = pos + wv; i < aend && wv < nteams; i += nteams) {
                const int me = sh.ord[i];
                Pol::subtract(P, S, G, A, plds, sh.c_t[me], sh.c_k[me], sh.c_c[me], lane, wv);
            }
            __syncthreads();
            HSCMP_STAMP(SB + 4);                                                  // subtraction
            // ---- P5: local re-correlation of the touched rows from the final residual (:1120, :1018-1051)
            {
                const int upa = Pol::units_per_atom(P);
                const int nu = (aend - pos) * upa;
                for (int u = wv; u < nu && wv < nteams; u += nteams) {
                    const int a = u / upa, q = u - a * upa;
                    const int me = sh.ord[pos + a];
                    const int p = sh.c_t[me];
                    Pol::recorrelate(P, S, G, A, plds, p, sh.c_k[me], q, (sh.c_flag[me] & RPF_INTERIOR) != 0, lane, wv);
                }
            }
            __syncthreads();
            HSCMP_STAMP(SB + 5);                                                  // re-correlation
            // (edge history of atoms at a signal end: read next by wave 0 -- in order behind this -- or behind a later barrier)
            if ((sh.nedge > 0 || !sh.spaced) && tid == 0)
                for (int i = pos; i < aend; ++i) {
                    const int me = sh.ord[i];
                    if (!(sh.c_flag[me] & RPF_INTERIOR)) Pol::after_atom(P, A, plds, sh.c_t[me]);
                }
            if (sh.converged) break;
            pos = gend;
        }
        if (full) {
            if (tid == 0) { sh.converged = 1; sh.stop = STOP_CAPACITY; sh.full = 0; }
            break;
        }
        // =========================== P6: segment maxima, slow stop rules (:1145-1163) ===========================
        if (n > 0 && !sh.converged) {
            // the segments that the rows of the applied atoms lie in, one wave per atom (its rows touch two segments as a
            // rule: both are fetched in one batch of loads); a segment that two atoms share is scanned twice, with the same result
            const int napplied = sh.napply;                                  // (the groups of a round are consecutive: ord[0, napply))
            for (int a = wv; a < napplied; a += kRpWaves) {
                const int p = sh.c_t[sh.ord[a]];
                const int lo = max(0, p - (W - 1)), hi = min(T - 1, p + (W - 1));
                const int sg0 = lo >> P.seg_shift, sg1 = hi >> P.seg_shift;
                if (P.seg <= 128 && sg1 == sg0 + 1) rp_scan_segment_pair<Pol::kScoreOnly>(P, G, wts, sh, sg0, lane);
                else for (int sg = sg0; sg <= sg1; ++sg) rp_scan_segment<Pol::kScoreOnly>(P, G, wts, sh, sg, lane);
            }
        }
        HSCMP_STAMP(SB + 12);
        if (tid == 0) {
            if (n == 0) { sh.converged = 1; if (sh.stop == STOP_RUNNING) sh.stop = STOP_EMPTY; }     // :1150-1153
            sh.rounds += 1;
            sh.offset = !sh.offset;
        }
        __syncthreads();
        HSCMP_STAMP(SB + 6);                                                      // segment maxima + round end
#ifdef HSCMP_DBG_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) { g_stamps[SB + 14] += 1; g_stamps[SB + 15] += (unsigned long long)n; }
#endif
        if (sh.converged) break;
    }

    __syncthreads();
    Pol::epilogue(P, S, A, plds, b);
    if (tid == 0) {
        stats[ST_NNZ] = sh.nnz; stats[ST_DUP] = sh.ndup; stats[ST_ROUNDS] = sh.rounds; stats[ST_STOP] = sh.stop;
        stats[ST_ITERS] = sh.iters; stats[ST_EVENTS] = sh.nev; stats[ST_SLOTS] = sh.nslots; stats[ST_OFFSET] = sh.offset;
        S.energy[2 * b + 1] = sh.e_res;
    }
}

// ------------------------------------------------------------------------------------------------
// Policy for single-feature float32 signals on the matrix cores (the score-only state of hscmp_mfma.h):
// every piece is the work of ONE wave with a strip of LDS of its own.
// LDS behind the control block: [dictionary image | weights][edge words][per wave: resolve window | tile window]
// ------------------------------------------------------------------------------------------------
template <int S4C, bool HAS_W> struct RpMfma {
    static_assert(S4C > 0, "compile-time chunk count only");
    using R = float;
    using Tile = TileF32;
    static constexpr int kMaxSegments = kMfmaMaxSeg;
    static constexpr int kMaxSel = 512;
    static constexpr bool kScoreOnly = true;
    static constexpr int TP = 32;
    using Shared = RpShared<float, kMfmaMaxSeg, kMaxSel>;
    using Args = MfmaArgs;
    static constexpr int kRw = 8 * S4C;                  // resolve window: W taps, zero padded
    static constexpr int kWin = TP + 8 * S4C + 32;       // tile window: 32 positions + taps (+ slack of the kk offset)

    struct Layout { float* dimg; float* wts; unsigned long long* edge; float* rw; float* win; };
    static __host__ __device__ size_t policy_lds_bytes(const Args& A)
    {
        return ((size_t)A.G * S4C * 256 + (HAS_W ? 32 * A.G : 0)) * sizeof(float) + kEdgeWords * sizeof(unsigned long long) +
               (size_t)kRpWaves * (kRw + kWin) * sizeof(float);
    }
    static size_t total_lds_bytes(const Args& A) { return ((sizeof(Shared) + 15) / 16) * 16 + policy_lds_bytes(A); }
    static __device__ __forceinline__ Layout layout(const Args& A, char* lds)
    {
        Layout L;
        L.dimg = reinterpret_cast<float*>(lds);
        L.wts = L.dimg + A.G * S4C * 256;
        L.edge = reinterpret_cast<unsigned long long*>(L.wts + (HAS_W ? 32 * A.G : 0));
        L.rw = reinterpret_cast<float*>(L.edge + kEdgeWords);
        L.win = L.rw + kRpWaves * kRw;
        return L;
    }
    static __device__ __forceinline__ const R* weights(const DevParams&, const State<R>& S, const Args&, char*) { return S.weights; }
    static __device__ __forceinline__ int units_per_atom(const DevParams& P) { return (2 * P.W - 1 + TP - 1) / TP; }
    static __device__ __forceinline__ int teams(const Args&) { return kRpWaves; }

    static __device__ __forceinline__ void prologue(const DevParams& P, const State<R>& S, const Sig<R>&, const Args& A, char* lds, int b)
    {
        const Layout L = layout(A, lds);
        lds_copy16(L.dimg, A.dimg, A.G * S4C * 256 * (int)sizeof(float), (int)threadIdx.x, kRpThreads);
        if (HAS_W) for (int i = threadIdx.x; i < 32 * A.G; i += kRpThreads) L.wts[i] = i < P.K ? S.weights[i] : 0.0f;
        for (int i = threadIdx.x; i < kRpWaves * (kRw + kWin); i += kRpThreads) L.rw[i] = 0.0f;      // padded taps stay zero
        if (threadIdx.x < kEdgeWords) L.edge[threadIdx.x] = S.edge[kEdgeWords * b + threadIdx.x];
        __syncthreads();
    }
    static __device__ __forceinline__ void epilogue(const DevParams&, const State<R>& S, const Args& A, char* lds, int b)
    {
        const Layout L = layout(A, lds);
        if (threadIdx.x < kEdgeWords) S.edge[kEdgeWords * b + threadIdx.x] = L.edge[threadIdx.x];
    }

    // the edge quirks of DESIGN.md section 2 make an atom order dependent: such atoms are applied alone
    static __device__ __forceinline__ bool is_interior(const DevParams& P, int p)
    {
        const int tstart = p - P.off - (P.W - 1), tend = p + P.W / 2 + (P.W - 1);
        return tstart >= 0 && tend <= P.T - 1 && !(!(P.W & 1) && p == P.T - 1 - P.W);
    }

    // local energies (:1002-1005) of the atom (k, c) at t from the window in `rw` (rw[w] = sample t - off + w): the pinned
    // tree over 256 strided partial sums, of which the window fills the first len <= 128 with one square each
    static __device__ __forceinline__ void window_energies(const DevParams& P, const Layout& L, const float* rw, int t, int k, float c,
                                                           int lane, float& eb, float& ea)
    {
        int s, e, es;
        const int len = centered_span(P.T, P.W, t, s, e, es);
        const float nc = -c;
        float pb[2] = {0.0f, 0.0f}, pa[2] = {0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = lane + 64 * u;
            if (q < len) {
                const float v = rw[es + q];
                const float prod = nc * L.dimg[dimg_index(k, es + q, S4C)];      // -c*D[k] rounded, then += (utils.py:120,129)
                const float vn = v + prod;
                pb[u] = v * v; pa[u] = vn * vn;
            }
        }
        wave_tree_down2(pb[0], pa[0]);
        wave_tree_down2(pb[1], pa[1]);
        const float b01 = pb[0] + pb[1], a01 = pa[0] + pa[1];                    // (P0 + P1) + (P2 + P3), the last two +0
        eb = b01 + 0.0f; ea = a01 + 0.0f;
    }

    static __device__ __forceinline__ void load_window(const DevParams& P, const Sig<R>& G, const Layout& L, float* rw, int t, int lane)
    {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float wres[2] = {0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 2; ++u) if (lane + 64 * u < P.W) wres[u] = edge_window_value(G.r, P.T, t - P.off + lane + 64 * u, t, L.edge);
#pragma unroll
        for (int u = 0; u < 2; ++u) if (lane + 64 * u < P.W) rw[lane + 64 * u] = wres[u];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    // (k, c) of position t (:935-946) from its group hint, local energies of that atom; all by one wave
    static __device__ __forceinline__ void candidate(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A, char* lds,
                                                     int t, int lane, int wv, int& k_out, R& c_out, R& eb, R& ea, int& flag)
    {
        const Layout L = layout(A, lds);
        float* rw = L.rw + wv * kRw;
        const int gh = G.bk[t];                                 // the hint travels with the window's samples
        load_window(P, G, L, rw, t, lane);
        const int g = __builtin_amdgcn_readfirstlane(gh);
        const int k = 32 * g + lane;
        Cand<R> best; best.s = -1.0f; best.i = INT_MAX;
        float bc = 0.0f;
        if (lane < 32 && k < P.K) {
            bc = resolve_chain<S4C>(L.dimg, rw, k, S4C);
            if (HAS_W) { const float sw = bc * L.wts[k]; best.s = fabsf(sw); } else best.s = fabsf(bc);
            best.i = k;
        }
        best = wave_argmax_first(best);                         // (lane l holds atom 32 g + l: lanes in index order)
        k_out = __builtin_amdgcn_readfirstlane(best.i);
        c_out = wave_bcast(bc, best.i & 31);
        window_energies(P, L, rw, t, k_out, c_out, lane, eb, ea);
        flag = is_interior(P, t) ? RPF_INTERIOR : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    static __device__ __forceinline__ void energies(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A, char* lds,
                                                    int t, int k, R c, int lane, int wv, R& eb, R& ea)
    {
        const Layout L = layout(A, lds);
        float* rw = L.rw + wv * kRw;
        load_window(P, G, L, rw, t, lane);
        window_energies(P, L, rw, t, k, c, lane, eb, ea);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    static __device__ __forceinline__ void subtract(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A, char* lds,
                                                    int p, int k, R c, int lane, int)
    {
        const Layout L = layout(A, lds);
        const int T = P.T, W = P.W;
        int s, e, es;
        const int len = centered_span(T, W, p, s, e, es);
        const float nc = -c;
        float v[2] = {0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 2; ++u) if (lane + 64 * u < len) v[u] = G.r[s + lane + 64 * u];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = lane + 64 * u;
            if (q < len) {
                const int m = s + q;
                const float prod = nc * L.dimg[dimg_index(k, es + q, S4C)];
                const float vn = v[u] + prod;
                // even W: an atom at T-1-W changes the sample that row T-1 reads through the reflection without
                // re-correlating that row -- keep what row T-1 saw (edge_window_value, hscmp_mfma.h)
                if (!(W & 1) && p == T - 1 - W && m == T - 1 - W / 2 && (L.edge[1] & 1ull) && L.edge[2] == 0ull) {
                    L.edge[3] = edge_bits_of(v[u]);
                    L.edge[2] = (unsigned long long)(m + 1);
                }
                G.r[m] = vn;
            }
        }
    }

    // rows 32 q .. 32 q + 31 of the atom's 2W-1 touched rows, from the (final) residual
    static __device__ __forceinline__ void recorrelate(const DevParams& P, const State<R>&, const Sig<R>& G, const Args& A, char* lds,
                                                       int p, int, int q, bool interior, int lane, int wv)
    {
        const Layout L = layout(A, lds);
        const int T = P.T, W = P.W;
        float* win = L.win + wv * kWin;
        const int nrows = 2 * W - 1, span = 3 * W - 2;
        const int tstart = p - P.off - (W - 1);                 // :1028-1033
        const int tend = p + W / 2 + (W - 1);                   // :1038
        const int sidx = tstart < 0 ? 0 : tstart;               // :1034
        const int eidx = tend > T - 1 ? T - 1 : tend;           // :1039
        const int nslice = eidx - sidx + 1;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        constexpr int kU = (TP + 8 * S4C + 63) / 64;
        float v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = lane + 64 * u, jj = TP * q + i;
            v[u] = 0.0f;
            if (i < TP + 8 * S4C && jj < span)                  // np.pad 'reflect', :1046; behind the span: zeros
                v[u] = G.r[interior ? tstart + jj : reflect_index(tstart + jj, sidx, nslice)];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int i = lane + 64 * u; if (i < TP + 8 * S4C) win[i] = v[u]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int grp;
        const float sc = mfma_tile_score_lean<S4C, HAS_W>(L.dimg, win, L.wts, A.G, lane, grp);
        const int row = TP * q + lane, t = p - (W - 1) + row;
        if (lane < TP && row < nrows && t >= 0 && t < T) {      // overlapReplace clipping (utils.py:133-161)
            G.bc[t] = sc;
            G.bk[t] = grp;                                      // the group hint of the row
        }
    }

    // rows lo..hi of an atom near a signal end now carry reflect-padded values (edge_window_value); one thread
    static __device__ __forceinline__ void after_atom(const DevParams& P, const Args& A, char* lds, int p)
    {
        const Layout L = layout(A, lds);
        const int T = P.T, W = P.W;
        const int lo = max(0, p - (W - 1)), hi = min(T - 1, p + (W - 1));
        if (lo < P.off) L.edge[0] |= bit_range(lo, min(hi, P.off - 1));
        const int rt0 = T - (W - 1 - P.off);                    // first position whose window passes T-1
        if (hi >= rt0) L.edge[1] |= bit_range(T - 1 - hi, T - 1 - max(lo, rt0));
        if (hi >= T - 1) L.edge[2] = 0ull;                      // row T-1 re-correlated: nothing stale any more
    }
};

// host-side dispatch -----------------------------------------------------------------------------
// can the round-parallel loop run these parameters at all (the policies add their own shape tests)
inline bool rp_params_ok(const DevParams& P, int maxsel_cap)
{
    return P.blocked && !P.has_scale && !P.select_only && P.maxsel <= maxsel_cap;
}

template <int S4C, bool HAS_W>
static int rp_mfma_launch_t(hipStream_t stream, const DevParams& P0, const State<float>& S, const MfmaArgs& A, bool dry)
{
    using Pol = RpMfma<S4C, HAS_W>;
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    const size_t lds = Pol::total_lds_bytes(A);
    if (lds > (size_t)160 * 1024) return -1;
    auto kern = iterate_rp_kernel<float, Pol>;
    if (set_dyn_lds((const void*)kern, lds) != hipSuccess) return -1;
    if (dry) return 0;
    hipLaunchKernelGGL(kern, dim3(P.B), dim3(kRpThreads), lds, stream, P, S, A);
    return 0;
}

// 0: launched (or, dry, could be); -1: this shape has no round-parallel form
inline int rp_mfma_launch(hipStream_t stream, const DevParams& P, const State<float>& S, const float* dimg, bool dry = false)
{
    if (!rp_params_ok(P, 512) || P.F != 1 || P.T < 3 * P.W - 2) return -1;
    const MfmaArgs A = mfma_args<float>(P, S, dimg);
    const bool hw = A.has_w != 0;
    switch (A.S4) {
    case 8: return hw ? rp_mfma_launch_t<8, true>(stream, P, S, A, dry) : rp_mfma_launch_t<8, false>(stream, P, S, A, dry);
    case 4: return hw ? rp_mfma_launch_t<4, true>(stream, P, S, A, dry) : rp_mfma_launch_t<4, false>(stream, P, S, A, dry);
    case 2: return hw ? rp_mfma_launch_t<2, true>(stream, P, S, A, dry) : rp_mfma_launch_t<2, false>(stream, P, S, A, dry);
    default: return -1;
    }
}

}  // namespace hscmp
