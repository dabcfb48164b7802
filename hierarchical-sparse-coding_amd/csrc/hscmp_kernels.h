// hscmp_kernels.h -- the kernels of the engine, generic over element type and shape.
//
//   prepare_kernel      residual = copy(x), signal energy            (modeling.py:1070-1072)
//   corr_init_kernel    initial zero-padded 'same' correlation       (modeling.py:1077, 149-188)
//                       reduced on the fly to the per-position best (or the full table for
//                       the convolve1d entry point)
//   iterate_kernel      the greedy loop                              (modeling.py:1086-1163)
//                       one persistent workgroup per signal: select (single or blocked arg-max,
//                       :899-982) -> coefficient bookkeeping (:1106-1114) -> residual subtract
//                       with energy tracking (:996-1016) -> local re-correlation of the 2W-1
//                       touched rows (:1018-1051) -> stop rules (:1125-1158), no host round trip.
//
// The re-correlation of the touched rows is a policy (`Recorr`): GenericRecorr below is the
// any-shape / any-dtype VALU version; hscmp_mfma.h provides the MFMA (matrix-core) one.
#pragma once

#include "hscmp_device.h"

#include <limits.h>

namespace hscmp {

// ------------------------------------------------------------------------------------------------
// prepare: residual <- x, energies, counters          grid = B, block = kThreads
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(kThreads) void prepare_kernel(DevParams P, State<R> S, const R* __restrict__ x)
{
    __shared__ R red[2 * kWaves];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t n = (int64_t)P.T * P.F;
    const R* xs = x + (int64_t)b * n;
    R* r = S.residual + (int64_t)b * n;
    R p = (R)0, q = (R)0;
    for (int64_t i = tid; i < n; i += kThreads) {   // strided partials, sequential in i (pinned order)
        const R v = xs[i];
        r[i] = v;
        const R sq = v * v;
        p = p + sq;
    }
    pinned_tree2(p, q, red);
    if (tid == 0) {
        S.energy[2 * b + 0] = p;
        S.energy[2 * b + 1] = p;
        int* st = S.stats + (int64_t)b * ST_COUNT;
        for (int i = 0; i < ST_COUNT; ++i) st[i] = 0;
        for (int i = 0; i < kEdgeWords; ++i) S.edge[kEdgeWords * b + i] = 0ull;
    }
}

// prepare for an input that was scattered from the previous level's coefficient slots straight into the
// (zero-filled) residual: the signal energy comes from the slots.  The pinned energy order is 256 strided
// partial sums, each sequential in the cell index i = t*F + f; a zero cell adds +0 and changes nothing, so
// partial (i mod 256) only needs its non-zero cells in ascending i.  Every thread collects its cells from the
// slot list and insertion-sorts them; if some partial has more cells than fit, the dense pass runs instead.
//   grid = count, block = kThreads
template <typename R>
__global__ __launch_bounds__(kThreads) void prepare_from_slots_kernel(DevParams P, State<R> S, const int* __restrict__ slot_t,
                                                                      const int* __restrict__ slot_k, const double* __restrict__ slot_a,
                                                                      const int* __restrict__ pstats, int pcap, int first, int has_min,
                                                                      double minc, const int* __restrict__ rl_cnt,
                                                                      const int* __restrict__ rl_f, int rl_cap)
{
    __shared__ R red[2 * kWaves];
    __shared__ int overflow;
    constexpr int kLocal = 96;
    const int b = blockIdx.x, tid = threadIdx.x, src = first + b;
    const int n = pstats[(int64_t)src * ST_COUNT + ST_SLOTS];
    long long cell[kLocal];
    R val[kLocal];
    int cnt = 0;
    bool over = false;
    if (tid == 0) overflow = 0;
    __syncthreads();
    // bound to overflow: go to the row lists at once.  With many workgroups in flight the walk (latency-bound, about as
    // long per workgroup whatever n) also beats the private lists much earlier (their insertion sort lives in scratch
    // memory: 21 vs 11 ms for 1024 signals of 8.5 k cells; 64 signals: 0.3 ms the other way round).
    const bool crowded = rl_cnt && rl_cap == 8 && (n > 48 * kThreads || (n > 16 * kThreads && gridDim.x > 512));
    if (crowded && tid == 0) overflow = 1;
    for (int i = 0; i < (crowded ? 0 : n); ++i) {   // (all threads read the same entries: broadcast loads)
        const double a = slot_a[(int64_t)src * pcap + i];
        if (a == 0.0 || (has_min && !(fabs(a) >= minc))) continue;
        const long long c = (long long)slot_t[(int64_t)src * pcap + i] * P.F + slot_k[(int64_t)src * pcap + i];
        if ((int)(c & (kThreads - 1)) != tid) continue;
        if (cnt == kLocal) { over = true; continue; }
        int j = cnt++;
        while (j > 0 && cell[j - 1] > c) { cell[j] = cell[j - 1]; val[j] = val[j - 1]; --j; }
        cell[j] = c; val[j] = (R)a;
    }
    if (over) atomicOr(&overflow, 1);
    __syncthreads();
    R p = (R)0, q = (R)0;
    if (overflow && rl_cnt && rl_cap == 8) {
        // too many cells per partial sum for the private lists: walk the per-row feature lists instead (rows ascending,
        // features ascending inside a row = ascending cell index).  Per batch of 64 rows every lane fetches ITS row's
        // list and cell values (coalesced, independent loads), then the rows are replayed to all lanes with shuffles and
        // each thread keeps the cells of its own partial sum; all four waves do the same walk.
        const int T = P.T, F = P.F, lane = tid & 63;
        const int* cnt = rl_cnt + (int64_t)b * T;
        const int* lf = rl_f + (int64_t)b * T * 8;
        const R* r = S.residual + (int64_t)b * T * F;
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int tl = t0 + lane;
            const int nl = tl < T ? cnt[tl] : 0;
            int fs[8];
            R vs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { fs[u] = INT_MAX; vs[u] = (R)0; }
            if (nl > 0 && nl <= 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int fv = lf[(int64_t)tl * 8 + u]; fs[u] = fv < 0 ? INT_MAX : fv; }
#pragma unroll
                for (int a = 0; a < 8; ++a)                  // ascending features (selection sort in registers)
#pragma unroll
                    for (int c2 = a + 1; c2 < 8; ++c2) if (fs[c2] < fs[a]) { const int tmp = fs[a]; fs[a] = fs[c2]; fs[c2] = tmp; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (fs[u] != INT_MAX) vs[u] = r[(int64_t)tl * F + fs[u]];
            }
            unsigned long long rows = __ballot(nl > 0);
            while (rows) {                                   // uniform within the wave
                const int q = __ffsll((long long)rows) - 1;
                rows &= rows - 1ull;
                const int t = t0 + q, n = __shfl(nl, q);
                if (n > 8) {                                 // overflowed list: the row is dense
                    for (int f = (int)(((unsigned)tid - (unsigned)((int64_t)t * F)) & (kThreads - 1)); f < F; f += kThreads) {
                        const R v = r[(int64_t)t * F + f]; const R sq = v * v; p = p + sq;
                    }
                    continue;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int f = __shfl(fs[u], q);
                    const R v = __shfl(vs[u], q);
                    if (f != INT_MAX && (int)(((int64_t)t * F + f) & (kThreads - 1)) == tid) { const R sq = v * v; p = p + sq; }
                }
            }
        }
    } else if (overflow) {
        const int64_t total = (int64_t)P.T * P.F;
        const R* r = S.residual + (int64_t)b * total;
        for (int64_t i = tid; i < total; i += kThreads) { const R v = r[i]; const R sq = v * v; p = p + sq; }
    } else {
        for (int j = 0; j < cnt; ++j) { const R sq = val[j] * val[j]; p = p + sq; }
    }
    pinned_tree2(p, q, red);
    if (tid == 0) {
        S.energy[2 * b + 0] = p;
        S.energy[2 * b + 1] = p;
        int* st = S.stats + (int64_t)b * ST_COUNT;
        for (int i = 0; i < ST_COUNT; ++i) st[i] = 0;
        for (int i = 0; i < kEdgeWords; ++i) S.edge[kEdgeWords * b + i] = 0ull;
    }
}

// The same energy by a counting sort of the slots: partial sum q owns the cells with index = q mod 256 and adds them in
// ascending index.  One word per kept slot -- (cell index / 256) above the slot number, `ibits` bits -- is dealt to the
// bucket of its partial sum in LDS (histogram, prefix, scatter), thread q sorts its bucket (a few dozen words) and walks it.
// No pass over rows, no scan of the whole list per thread: 17 k slots per signal (BASELINE config 5) take tens of
// microseconds where the row-list walk of prepare_from_slots_kernel takes 9 ms.
//   grid = count, block = kThreads, dynamic LDS = 4 * max slots;   requires cell index / 256 < 2^(32 - ibits), slots < 2^ibits
template <typename R>
__global__ __launch_bounds__(kThreads) void prepare_from_slots_sorted_kernel(DevParams P, State<R> S, const int* __restrict__ slot_t,
                                                                             const int* __restrict__ slot_k, const double* __restrict__ slot_a,
                                                                             const int* __restrict__ pstats, int pcap, int first, int has_min,
                                                                             double minc, int ibits)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* words = reinterpret_cast<unsigned*>(smem);
    __shared__ R red[2 * kWaves];
    __shared__ int cnt[kThreads], start[kThreads + 1];
    const int b = blockIdx.x, tid = threadIdx.x, src = first + b;
    const int n = pstats[(int64_t)src * ST_COUNT + ST_SLOTS];
    const int* st = slot_t + (int64_t)src * pcap;
    const int* sk = slot_k + (int64_t)src * pcap;
    const double* sa = slot_a + (int64_t)src * pcap;
    auto kept = [&](double a) { return !(a == 0.0 || (has_min && !(fabs(a) >= minc))); };
    cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kThreads)
        if (kept(sa[i])) atomicAdd(&cnt[(int)(((long long)st[i] * P.F + sk[i]) & (kThreads - 1))], 1);
    __syncthreads();
    if (tid < 64) {                                      // exclusive prefix over the 256 buckets by one wave
        int c4[4], local = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { c4[u] = cnt[4 * tid + u]; local += c4[u]; }
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (tid >= d) incl += o; }
        int run = incl - local;
#pragma unroll
        for (int u = 0; u < 4; ++u) { start[4 * tid + u] = run; run += c4[u]; }
        if (tid == 63) start[kThreads] = incl;
    }
    __syncthreads();
    cnt[tid] = 0;                                        // cursors
    __syncthreads();
    for (int i = tid; i < n; i += kThreads)
        if (kept(sa[i])) {
            const long long c = (long long)st[i] * P.F + sk[i];
            const int q = (int)(c & (kThreads - 1));
            words[start[q] + atomicAdd(&cnt[q], 1)] = ((unsigned)(c >> 8) << ibits) | (unsigned)i;
        }
    __syncthreads();
    // every bucket in ascending cell index (the words are distinct).  The buckets are as uneven as the atoms of the level
    // below are popular -- cell index mod 256 is the atom number when that level has 128 or 256 atoms -- so a bucket is
    // sorted by a WAVE: every lane counts the smaller words for its entries (LDS broadcast reads), then all write.
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int q = wv; q < kThreads; q += kWaves) {
            const int s0 = start[q], m = start[q + 1] - s0;
            if (m < 2) continue;                                             // (uniform)
            if (m > 16 * 64) {                                               // beyond the registers of a wave: one lane, insertion sort
                if (lane == 0)
                    for (int i = s0 + 1; i < s0 + m; ++i) {
                        const unsigned w = words[i];
                        int j = i;
                        while (j > s0 && words[j - 1] > w) { words[j] = words[j - 1]; --j; }
                        words[j] = w;
                    }
                continue;
            }
            unsigned mine[16];
            int rank[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { mine[u] = lane + 64 * u < m ? words[s0 + lane + 64 * u] : 0xffffffffu; rank[u] = 0; }
            for (int j0 = 0; j0 < m; j0 += 4) {
                unsigned o[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) o[v] = j0 + v < m ? words[s0 + j0 + v] : 0xffffffffu;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (64 * u < m) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) rank[u] += o[v] < mine[u] ? 1 : 0;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 16; ++u) if (lane + 64 * u < m) words[s0 + rank[u]] = mine[u];
        }
    }
    __syncthreads();
    const int b0 = start[tid], b1 = start[tid + 1];
    R p = (R)0, q2 = (R)0;
    const unsigned imask = (1u << ibits) - 1u;
    constexpr int kU = 8;                                // loads of a batch issued together
    for (int i0 = b0; i0 < b1; i0 += kU) {
        double a[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) a[u] = i0 + u < b1 ? sa[words[i0 + u] & imask] : 0.0;
#pragma unroll
        for (int u = 0; u < kU; ++u) if (i0 + u < b1) { const R v = (R)a[u]; const R sq = v * v; p = p + sq; }
    }
    pinned_tree2(p, q2, red);
    if (tid == 0) {
        S.energy[2 * b + 0] = p;
        S.energy[2 * b + 1] = p;
        int* stt = S.stats + (int64_t)b * ST_COUNT;
        for (int i = 0; i < ST_COUNT; ++i) stt[i] = 0;
        for (int i = 0; i < kEdgeWords; ++i) S.edge[kEdgeWords * b + i] = 0ull;
    }
}

// ------------------------------------------------------------------------------------------------
// generic initial correlation: one thread per output position, all atoms, pinned fma chain
//   grid = (ceil(Tout/kThreads), B), block = kThreads
//   TABLE = false: write per-position best (best_c, best_k);  TABLE = true: write out[Tout][K]
// ------------------------------------------------------------------------------------------------
template <typename R, bool TABLE>
__global__ __launch_bounds__(kThreads) void corr_init_generic_kernel(DevParams P, State<R> S, const R* __restrict__ src,
                                                                      int lead, int Tout, R* __restrict__ table)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= Tout) return;
    const int T = P.T, K = P.K, W = P.W, F = P.F;
    const R* x = src + (int64_t)b * T * F;
    const R* __restrict__ D = S.D;
    R bs = (R)-1, bc = (R)0;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
        R acc = (R)0;
        const R* dk = D + (int64_t)k * W * F;
        for (int f = 0; f < F; ++f)
            for (int w = 0; w < W; ++w) {
                const int g = t - lead + w;
                const R xv = (g >= 0 && g < T) ? x[(int64_t)g * F + f] : (R)0;
                acc = rfma(xv, dk[w * F + f], acc);
            }
        if (TABLE) {
            table[((int64_t)b * Tout + t) * K + k] = acc;
        } else {
            const R sc = score_of(acc, k, S.weights);
            if (sc > bs) { bs = sc; bc = acc; bk = k; }
        }
    }
    if (!TABLE) {
        S.best_c[(int64_t)b * T + t] = bc;
        S.best_k[(int64_t)b * T + t] = bk;
    }
}

// Rows of the inner-product table of a MULTI-FEATURE input (hierarchical levels >= 1: x [T][F] holds the previous level's
// coefficients and is almost all zeros), for LoCOMP's device-resident table.  One workgroup per output row t:
//   1. the W x F cells of the row's window are visited in the pinned chain order (f outer, w inner) and the non-zero ones
//      are compacted, in that order, into an LDS list -- a zero factor contributes fma(0, d, acc) = acc exactly;
//   2. thread k runs its chain over the list only: table[t][k] = sum x * D[k][w][f].
// The dense form (corr_init_generic_kernel / update_rows_kernel) forms W*F products per output -- 4352 at config-4 level
// 1 -- of which ~27 are non-zero.  Rows whose window holds more non-zeros than the list fall back to the dense chain.
//   p < 0:  rows row0 .. row0+nrows-1 with ZERO padding ('same' initial correlation, modeling.py:159-164), lead = off
//   p >= 0: rows p-(W-1) .. p+(W-1), REFLECT padding w.r.t. the slice of the update (modeling.py:1018-1051); rows
//           outside [0, T) are skipped (overlapReplace clipping)
//   grid = number of rows, block = kThreads, dynamic LDS = cap * (sizeof(R) + 4)
template <typename R>
__global__ __launch_bounds__(kThreads) void table_rows_sparse_kernel(DevParams P, const R* __restrict__ r, const R* __restrict__ D, int row0, int p,
                                                                     R* __restrict__ table, int cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    R* lx = reinterpret_cast<R*>(smem);
    int* lwf = reinterpret_cast<int*>(lx + cap);
    __shared__ int wcount[kWaves];
    const int T = P.T, K = P.K, W = P.W, F = P.F, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool reflect = p >= 0;
    const int t = reflect ? p - (W - 1) + (int)blockIdx.x : row0 + (int)blockIdx.x;
    if (t < 0 || t >= T) return;                                   // (uniform)
    int sidx = 0, nslice = 1;
    if (reflect) {
        const int tstart = p - P.off - (W - 1), tend = p + W / 2 + (W - 1);
        sidx = tstart < 0 ? 0 : tstart;
        nslice = (tend > T - 1 ? T - 1 : tend) - sidx + 1;
    }
    // 1. ordered compaction, 256 cells per pass
    const int ncell = W * F;
    int base = 0;
    bool overflow = false;
    for (int c0 = 0; c0 < ncell; c0 += kThreads) {
        const int i = c0 + tid;
        R xv = (R)0;
        int wf = 0;
        if (i < ncell) {
            const int f = i / W, w = i - f * W;
            int g = t - P.off + w;
            if (reflect) { g = reflect_index(g, sidx, nslice); xv = r[(int64_t)g * F + f]; }
            else xv = (g >= 0 && g < T) ? r[(int64_t)g * F + f] : (R)0;
            wf = (w << 16) | f;
        }
        const unsigned long long mine = __ballot(xv != (R)0);
        if (lane == 0) wcount[wv] = __popcll(mine);
        __syncthreads();
        int before = base;
        for (int q = 0; q < wv; ++q) before += wcount[q];
        const int pos = before + __popcll(mine & ((1ull << lane) - 1ull));
        if (xv != (R)0) { if (pos < cap) { lx[pos] = xv; lwf[pos] = wf; } }
        int all = 0;
        for (int q = 0; q < kWaves; ++q) all += wcount[q];
        base += all;
        __syncthreads();
    }
    overflow = base > cap;                                          // (uniform)
    const int n = base;
    // 2. one chain per atom
    for (int k = tid; k < K; k += kThreads) {
        const R* dk = D + (int64_t)k * W * F;
        R acc = (R)0;
        if (!overflow) {
            for (int q = 0; q < n; ++q) { const int wf = lwf[q]; acc = rfma(lx[q], dk[(wf >> 16) * F + (wf & 0xffff)], acc); }
        } else {
            for (int f = 0; f < F; ++f)
                for (int w = 0; w < W; ++w) {
                    int g = t - P.off + w;
                    R xv;
                    if (reflect) { g = reflect_index(g, sidx, nslice); xv = r[(int64_t)g * F + f]; }
                    else xv = (g >= 0 && g < T) ? r[(int64_t)g * F + f] : (R)0;
                    acc = rfma(xv, dk[w * F + f], acc);
                }
        }
        table[(int64_t)t * K + k] = acc;
    }
}

// Window assignment of the convolutional k-means learner (modeling.py:454-460): for each of N windows
// [L][F] the 'valid' correlation with every atom (Tout = L-W+1 positions) and the flat arg-max of |c|
// over (position, atom) in C order (ties: lowest position, then lowest atom).
//   grid = N, block = kThreads; the window is staged in LDS when it fits `lds_elems` elements
template <typename R>
__global__ __launch_bounds__(kThreads) void assign_windows_kernel(const R* __restrict__ windows, int L, int K, int W, int F,
                                                                  const R* __restrict__ D, int lds_elems,
                                                                  int* __restrict__ out_t, int* __restrict__ out_k, R* __restrict__ out_c)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    R* win = reinterpret_cast<R*>(smem);
    __shared__ R red_c[kThreads];
    __shared__ int red_o[kThreads];
    const int n = blockIdx.x, tid = threadIdx.x;
    const R* x = windows + (int64_t)n * L * F;
    const bool staged = L * F <= lds_elems;
    if (staged) {
        for (int i = tid; i < L * F; i += kThreads) win[i] = x[i];
        __syncthreads();
    }
    const R* src = staged ? win : x;
    const int Tout = L - W + 1;
    R bs = (R)-1, bc = (R)0;
    int bo = INT_MAX;
    for (int o = tid; o < Tout * K; o += kThreads) {           // ascending o per thread: '>' keeps the first of equals
        const int t = o / K, k = o - t * K;
        const R* dk = D + (int64_t)k * W * F;
        R acc = (R)0;
        for (int f = 0; f < F; ++f)
            for (int w = 0; w < W; ++w) acc = rfma(src[(t + w) * F + f], dk[w * F + f], acc);    // pinned chain: f outer, w inner
        const R sc = rabs(acc);
        if (sc > bs) { bs = sc; bc = acc; bo = o; }
    }
    red_c[tid] = bc; red_o[tid] = bo;
    __syncthreads();
    for (int stride = kThreads / 2; stride > 0; stride >>= 1) {
        if (tid < stride) {
            const R ca = red_c[tid], cb = red_c[tid + stride];
            const int oa = red_o[tid], ob = red_o[tid + stride];
            const bool take = ob != INT_MAX && (oa == INT_MAX || rabs(cb) > rabs(ca) || (rabs(cb) == rabs(ca) && ob < oa));
            if (take) { red_c[tid] = cb; red_o[tid] = ob; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int o = red_o[0] == INT_MAX ? 0 : red_o[0];       // all-NaN window: numpy's argmax returns the first NaN; not reproduced
        out_t[n] = o / K; out_k[n] = o - (o / K) * K; out_c[n] = red_c[0];
    }
}

// dense next-level input from the previous level's coefficient slots (modeling.py:1489 `todense()` of the
// CSC matrix built by :1171-1181): x[b][t][k] = slot_a unless it is zero or below min_coefficients.
//   grid = count, block = kThreads; x [count][T][F] must be zero filled
template <typename R>
__global__ __launch_bounds__(kThreads) void scatter_slots_kernel(R* __restrict__ x, int T, int F, const int* __restrict__ slot_t,
                                                                 const int* __restrict__ slot_k, const double* __restrict__ slot_a,
                                                                 const int* __restrict__ stats, int cap, int first, int has_min, double minc,
                                                                 unsigned char* __restrict__ rowflag, int* __restrict__ rl_cnt,
                                                                 int* __restrict__ rl_f, int rl_cap)
{
    const int b = blockIdx.x, src = first + b;
    const int n = stats[(int64_t)src * ST_COUNT + ST_SLOTS];
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const double a = slot_a[(int64_t)src * cap + i];
        if (a == 0.0 || (has_min && !(fabs(a) >= minc))) continue;
        const int t = slot_t[(int64_t)src * cap + i];
        x[((int64_t)b * T + t) * F + slot_k[(int64_t)src * cap + i]] = (R)a;
        rowflag[(int64_t)b * T + t] = 1;                   // non-zero input row (rowflag must be zero filled)
        if (rl_cnt) {                                      // per-row list of the features that may be non-zero
            const int c = atomicAdd(&rl_cnt[(int64_t)b * T + t], 1);
            if (c < rl_cap) rl_f[((int64_t)b * T + t) * rl_cap + c] = slot_k[(int64_t)src * cap + i];
        }
    }
}

// Zero the cells of x [rows][F] that the row lists name (a row whose list overflowed: all of it).  After an encode
// whose loop kept the lists current these are the only cells of the dense level input / residual that can be
// non-zero, so the next batch's "zero filled" buffer costs a pass over the list counters instead of a memset of
// rows x F values (config 4, 1024 signals: 137 GB).    grid = ceil(rows / kThreads)
template <typename R>
__global__ __launch_bounds__(kThreads) void clear_listed_cells_kernel(R* __restrict__ x, int64_t rows, int F, const int* __restrict__ rl_cnt,
                                                                      const int* __restrict__ rl_f, int rl_cap)
{
    const int64_t row = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (row >= rows) return;
    const int n = rl_cnt[row];
    if (n <= 0) return;
    R* xr = x + row * F;
    if (n > rl_cap) { for (int f = 0; f < F; ++f) xr[f] = (R)0; return; }
    for (int i = 0; i < n; ++i) {
        const int f = rl_f[row * rl_cap + i];
        if (f >= 0 && f < F) xr[f] = (R)0;
    }
}

// Per-row feature lists of a dense multi-feature input x [B][T][F]: rl_cnt[b][t] = number of non-zero
// features of row t (may exceed rl_cap: the row is then treated as dense), rl_f[b][t][0..rl_cap) their
// indices in any order (-1 = empty; both arrays must be pre-filled with 0 / -1).
//   grid = (B, splits), block = kThreads
template <typename R>
__global__ __launch_bounds__(kThreads) void build_row_lists_kernel(const R* __restrict__ x, int T, int F, int* __restrict__ rl_cnt,
                                                                   int* __restrict__ rl_f, int rl_cap)
{
    const int b = blockIdx.x;
    const int64_t n = (int64_t)T * F;
    const R* xb = x + (int64_t)b * n;
    for (int64_t e = (int64_t)blockIdx.y * kThreads + threadIdx.x; e < n; e += (int64_t)gridDim.y * kThreads) {
        if (xb[e] != (R)0) {
            const int t = (int)(e / F), f = (int)(e - (int64_t)t * F);
            const int c = atomicAdd(&rl_cnt[(int64_t)b * T + t], 1);
            if (c < rl_cap) rl_f[((int64_t)b * T + t) * rl_cap + c] = f;
        }
    }
}

// per-position best of a materialised inner-product table ip[T][K] (entry point of _selectBestAtoms)
template <typename R>
__global__ __launch_bounds__(kThreads) void table_to_best_kernel(const R* __restrict__ ip, int T, int K, const R* __restrict__ w,
                                                                 R* __restrict__ bc, int* __restrict__ bk)
{
    const int t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= T) return;
    R bs = (R)-1, c = (R)0;
    int kk = 0;
    for (int k = 0; k < K; ++k) {
        const R v = ip[(int64_t)t * K + k];
        const R sc = score_of(v, k, w);
        if (sc > bs) { bs = sc; c = v; kk = k; }
    }
    bc[t] = c;
    bk[t] = kk;
}

// rows p-(W-1) .. p+(W-1) of the inner-product table from the reflect-padded residual span
// (modeling.py:1018-1051, entry point of _updateInnerProducts): out[(2W-1)][K], or -- `table` given -- straight into the
// rows of a device-resident table [T][K] with the clipping of overlapReplace (utils.py:133-161)
template <typename R>
__global__ __launch_bounds__(kThreads) void update_rows_kernel(DevParams P, const R* __restrict__ r, const R* __restrict__ D, int p,
                                                               R* __restrict__ out, R* __restrict__ table)
{
    const int T = P.T, K = P.K, W = P.W, F = P.F;
    const int tstart = p - P.off - (W - 1), tend = p + W / 2 + (W - 1);
    const int sidx = tstart < 0 ? 0 : tstart, eidx = tend > T - 1 ? T - 1 : tend, nslice = eidx - sidx + 1;
    const int nrows = 2 * W - 1;
    for (int o = blockIdx.x * kThreads + threadIdx.x; o < nrows * K; o += gridDim.x * kThreads) {
        const int row = o / K, k = o - row * K;
        const int t = p - (W - 1) + row;
        const R* dk = D + (int64_t)k * W * F;
        R acc = (R)0;
        for (int f = 0; f < F; ++f)
            for (int w = 0; w < W; ++w) {
                const int g = reflect_index(t - P.off + w, sidx, nslice);
                acc = rfma(r[(int64_t)g * F + f], dk[w * F + f], acc);
            }
        if (table) { if (t >= 0 && t < T) table[(int64_t)t * K + k] = acc; }
        else out[o] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// shared state of the greedy loop
// ------------------------------------------------------------------------------------------------
constexpr int kBloomWords = 256;          // 8192-bit Bloom filter over the selected (t,k) pairs
__device__ __forceinline__ unsigned bloom_hash(int t, int k)
{
    unsigned h = (unsigned)t * 2654435761u + (unsigned)k * 40503u;
    h ^= h >> 15;
    return h & (kBloomWords * 32 - 1);
}

constexpr int kSelLds = 16;               // rounds of at most this many candidates keep their selection lists in LDS

template <typename R, int MAXSEG, bool WITH_PART = true, bool WITH_CK = true> struct IterSharedT {
    unsigned bloom[WITH_CK ? kBloomWords : 1];    // step-by-step atom body: (t,k) pairs that own a slot (fused policies keep their own)
    R seg_score[MAXSEG];
    int seg_t[MAXSEG];
    R seg_c[WITH_CK ? MAXSEG : 1];            // coefficient / atom of the segment maximum (not kept by
    int seg_k[WITH_CK ? MAXSEG : 1];          // score-only policies, which resolve them on demand)
    R rseg[MAXSEG];           // max |residual| per segment (toleranceResidualScale only)
    R part_s[WITH_PART ? kThreads : 1];       // GenericRecorr's cross-group merge buffers
    R part_c[WITH_PART ? kThreads : 1];
    int part_k[WITH_PART ? kThreads : 1];
    R red[2 * kWaves];
    unsigned touched[(MAXSEG + 31) / 32];     // step-by-step body, blocked rounds: segments whose maxima are refreshed at the round end
    // blocked selection (:908-962): raw / ordered candidate lists of a round (longer lists live in State::sel_*)
    int sel_t[2 * kSelLds]; int sel_k[2 * kSelLds]; R sel_c[2 * kSelLds];
    Cand<R> cred[kWaves];
    int wtot[kWaves];
    unsigned bar;             // SoftSync: arrivals at the signal's barrier (several signals per workgroup)
    int bar_cnt;              // SoftSync::count scratch
    // control block (written by thread 0, read by all after a barrier)
    int nsel, converged, stop, found, skip;
    int nullsel;              // a fused atom body found the selected coefficient null (:974): the round selected nothing
    int hashed;               // the signal's slot hash table is built and answers the duplicate lookups (see slot_find)
    unsigned fpos;            // single arg-max rounds: where slot_find stopped (the free entry a new slot takes)
    int nnz, ndup, rounds, iters, nev, nslots, offset;
    int atom_t, atom_k;
    R atom_c;
    R e_sig, e_res;
    // LoCOMP, a selection committed by its wave alone (hscmp_locomp.h): 5, + 2 when the pursuit of this signal ends with it -- locomp_atom's
    // result, the ONLY shared word the other waves read behind the selection's one barrier.  Two words, taken in turn: a late wave may still
    // have to read the first selection's word when the second one's is written.  (Not in the group state: the waves' workspaces alias that.)
    int lc_flag[2];
    // fused atom bodies: {converged, events, slots, hashed} as of the end of the last atom, written by the bookkeeping
    // lane in front of the atom's last barrier.  Every wave fetches it with ONE 16-byte read behind that barrier and
    // carries it in scalar registers (FusedCtl): beside the matrix instructions of co-resident signals each LDS round trip
    // of a wave costs ~1000 cycles, and the control flow between two atoms used to make five of them, one after another.
    alignas(16) int ctl[4];
};

struct FusedCtl { int converged, nev, nslots, hashed; };
template <typename SH> __device__ __forceinline__ FusedCtl fused_ctl_fetch(const SH& sh)
{
    const int4 v = *reinterpret_cast<const int4*>(sh.ctl);
    FusedCtl c;
    c.converged = __builtin_amdgcn_readfirstlane(v.x); c.nev = __builtin_amdgcn_readfirstlane(v.y);
    c.nslots = __builtin_amdgcn_readfirstlane(v.z); c.hashed = __builtin_amdgcn_readfirstlane(v.w);
    return c;
}

template <typename R> struct Sig {   // per-signal views
    R* r; R* bc; int* bk;
    int* ev_t; int* ev_k; R* ev_c;
    int* slot_t; int* slot_k; double* slot_a;
    unsigned long long* hkey; int* hval;
    int* sel_t; int* sel_k; R* sel_c;
    int* head;          // (round-parallel loop only)
    double* lgram;      // (LoCOMP only: Gram matrix of a group too large for its LDS copy)
};

// ------------------------------------------------------------------------------------------------
// (t,k) -> coefficient slot (:1106-1114 looks the pair up in the dict of coefficients).  Short slot lists are
// searched directly behind a Bloom filter in LDS; from kSlotHashMin slots on the filter saturates and every atom
// would pay for a scan of the whole list, so the workgroup builds an open-addressing table in global memory
// (linear probing, load <= 1/2) and keeps it current.  All table accesses are agent-scope atomics (L2): the
// compare-and-swap inserts and the probing loads then see the same copy whatever the vector cache holds.
// ------------------------------------------------------------------------------------------------
constexpr int kSlotHashMin = 1024;
constexpr unsigned long long kSlotEmpty = ~0ull;
__device__ __forceinline__ unsigned long long slot_key(int t, int k) { return ((unsigned long long)(unsigned)t << 32) | (unsigned)k; }
__device__ __forceinline__ unsigned slot_hash(unsigned long long key)
{
    key *= 0x9E3779B97F4A7C15ull;
    return (unsigned)(key >> 32);
}
__device__ __forceinline__ unsigned long long hkey_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hkey_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int hval_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hval_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// slot of (t,k) or -1; pos = the entry that holds it / the free entry the probe ended on (one thread)
template <typename R> __device__ __forceinline__ int slot_find(const Sig<R>& G, unsigned hmask, int t, int k, unsigned& pos)
{
    const unsigned long long key = slot_key(t, k);
    unsigned h = slot_hash(key) & hmask;
    for (;;) {
        const unsigned long long q = hkey_load(G.hkey + h);
        if (q == key) { pos = h; return hval_load(G.hval + h); }
        if (q == kSlotEmpty) { pos = h; return -1; }
        h = (h + 1) & hmask;
    }
}
// (t,k) is not in the table: enter it (any number of threads at once)
template <typename R> __device__ __forceinline__ void slot_insert(const Sig<R>& G, unsigned hmask, int t, int k, int si)
{
    const unsigned long long key = slot_key(t, k);
    unsigned h = slot_hash(key) & hmask;
    for (;;) {
        unsigned long long expect = kSlotEmpty;
        if (__hip_atomic_compare_exchange_strong(G.hkey + h, &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        h = (h + 1) & hmask;
    }
    hval_store(G.hval + h, si);
}
// the entry `pos` is free (slot_find ended there and nothing was entered since): one store, nobody waits
template <typename R> __device__ __forceinline__ void slot_insert_at(const Sig<R>& G, unsigned pos, int t, int k, int si)
{
    hval_store(G.hval + pos, si);
    hkey_store(G.hkey + pos, slot_key(t, k));
}
// table of the first `nslots` slots, by the whole workgroup (the slot list must be visible: call after a full
// barrier; the caller ends with one as well)
template <typename R, typename SY> __device__ __forceinline__ void slot_table_build(const Sig<R>& G, unsigned hmask, int nslots, SY& sy)
{
    for (unsigned i = ltid(); i <= hmask; i += kThreads) hkey_store(G.hkey + i, kSlotEmpty);
    sy.full();
    for (int i = ltid(); i < nslots; i += kThreads) slot_insert(G, hmask, G.slot_t[i], G.slot_k[i], i);
}

// arg-max of the per-position best over positions [t0,t1) by one wave; result in all lanes.
// SO (score-only policy): G.bc[t] already IS the score max_k |c[t,k]*w_k|.
template <bool SO, typename R>
__device__ __forceinline__ Cand<R> wave_range_argmax(const Sig<R>& G, const R* w, int t0, int t1, int lane)
{
    Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
    for (int t = t0 + lane; t < t1; t += 64) {
        R s;
        if constexpr (SO) s = G.bc[t]; else s = score_of(G.bc[t], G.bk[t], w);
        if (s > best.s) { best.s = s; best.i = t; }
    }
    return wave_argmax(best);
}

template <bool SO, typename R, typename SH>
__device__ __forceinline__ void scan_segment(const DevParams& P, const Sig<R>& G, const R* w, SH& sh, int sg, int lane)
{
    const int t0 = (sg << P.seg_shift);
    const int t1 = min(P.T, t0 + P.seg);
    Cand<R> win = wave_range_argmax<SO>(G, w, t0, t1, lane);
    if (lane == 0) {
        if (win.i == INT_MAX) { win.i = t0; win.s = (R)0; }
        sh.seg_score[sg] = win.s;
        sh.seg_t[sg] = win.i;
        if constexpr (!SO) { sh.seg_c[sg] = G.bc[win.i]; sh.seg_k[sg] = G.bk[win.i]; }
    }
}

// scan_segment for a policy that still holds the rows it just re-correlated in LDS (rows [rt0, rt0+rn): atom rk,
// INT_MAX = the default winner k = 0 with coefficient r0, else coefficient rc): those positions are taken from LDS,
// the others from global memory, so the scan does not wait for the rows' global stores.  Same maxima.
template <typename R, typename SH>
__device__ __forceinline__ void scan_segment_rows(const DevParams& P, const Sig<R>& G, const R* w, SH& sh, int sg, int lane,
                                                  const int* rk, const R* rc, const R* r0, int rt0, int rn)
{
    const int t0 = (sg << P.seg_shift);
    const int t1 = min(P.T, t0 + P.seg);
    auto fetch = [&](int t, R& c, int& k) {
        const int row = t - rt0;
        if (row >= 0 && row < rn) { k = rk[row]; if (k == INT_MAX) { k = 0; c = r0[row]; } else c = rc[row]; }
        else { c = G.bc[t]; k = G.bk[t]; }
    };
    Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
    R bc = (R)0;
    int bk = 0;
    for (int t = t0 + lane; t < t1; t += 64) {
        R c; int k;
        fetch(t, c, k);
        const R sc = score_of(c, k, w);
        if (sc > best.s) { best.s = sc; best.i = t; bc = c; bk = k; }
    }
    Cand<R> win = wave_argmax(best);
    if (win.i == INT_MAX) {
        win.i = t0; win.s = (R)0;
        fetch(t0, bc, bk);                                   // (every lane: uniform)
    } else {
        const int owner = __ffsll((long long)__ballot(best.i == win.i)) - 1;
        bc = __shfl(bc, owner); bk = __shfl(bk, owner);
    }
    if (lane == 0) { sh.seg_score[sg] = win.s; sh.seg_t[sg] = win.i; sh.seg_c[sg] = bc; sh.seg_k[sg] = bk; }
}

// arg-max of the per-position best over the block [lo,hi) by one wave, through the segment maxima:
// segments that lie entirely inside the block are read from LDS, only the two ragged ends are
// scanned.  Same result as wave_range_argmax (maximum score, lowest position among equals).
template <bool SO, typename R, typename SH>
__device__ __forceinline__ Cand<R> wave_block_argmax(const DevParams& P, const Sig<R>& G, const R* w, const SH& sh,
                                                     int lo, int hi, int lane)
{
    const int sgA = (lo + P.seg - 1) >> P.seg_shift;                        // first segment entirely inside
    const int sgB = (hi >= P.T) ? P.nseg - 1 : (hi >> P.seg_shift) - 1;     // last one (the last segment of the signal may be short)
    if (sgA > sgB) return wave_range_argmax<SO>(G, w, lo, hi, lane);
    Cand<R> best; best.s = (R)-1; best.i = INT_MAX;
    const int headEnd = sgA << P.seg_shift, tailBegin = min(hi, (sgB + 1) << P.seg_shift);
    // per lane the candidates come in ascending position, so '>' keeps the first of equals
    for (int t = lo + lane; t < headEnd; t += 64) {
        R sc;
        if constexpr (SO) sc = G.bc[t]; else sc = score_of(G.bc[t], G.bk[t], w);
        if (sc > best.s) { best.s = sc; best.i = t; }
    }
    for (int sg = sgA + lane; sg <= sgB; sg += 64) {
        const R sc = sh.seg_score[sg];
        if (sc > best.s) { best.s = sc; best.i = sh.seg_t[sg]; }
    }
    for (int t = tailBegin + lane; t < hi; t += 64) {
        R sc;
        if constexpr (SO) sc = G.bc[t]; else sc = score_of(G.bc[t], G.bk[t], w);
        if (sc > best.s) { best.s = sc; best.i = t; }
    }
    return wave_argmax(best);
}

template <typename R, typename SH>
__device__ __forceinline__ void rscan_segment(const DevParams& P, const Sig<R>& G, SH& sh, int sg, int lane)
{
    const int64_t i0 = (int64_t)(sg << P.seg_shift) * P.F;
    const int64_t i1 = (int64_t)min(P.T, ((sg + 1) << P.seg_shift)) * P.F;
    R m = (R)0;
    for (int64_t i = i0 + lane; i < i1; i += 64) { const R a = rabs(G.r[i]); m = a > m ? a : m; }
    m = wave_max(m);
    if (lane == 0) sh.rseg[sg] = m;
}

// block-wide stable compaction of list entries [0,n) with predicate pred(i): src -> dst; returns count
template <typename R, typename SH, typename Pred, typename SY>
__device__ __forceinline__ int block_compact(int n, Pred pred, const int* st, const int* sk, const R* sc,
                                              int* dt, int* dk, R* dc, SH& sh, SY& sy)
{
    const int tid = ltid(), lane = tid & 63, wv = tid >> 6;
    int running = 0;
    for (int base = 0; base < n; base += kThreads) {
        const int i = base + tid;
        const int f = (i < n) ? (pred(i) ? 1 : 0) : 0;
        const unsigned long long mask = __ballot(f);
        const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
        sy.full();
        if (lane == 0) sh.wtot[wv] = __popcll(mask);
        sy.full();
        int before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < kWaves; ++q) { const int c = sh.wtot[q]; if (q < wv) before += c; total += c; }
        if (f) {
            const int o = running + before + prefix;
            dt[o] = st[i]; dk[o] = sk[i]; dc[o] = sc[i];
        }
        running += total;
    }
    sy.full();
    return running;
}

// energy of the clipped window centred at t, pinned order; result valid in thread 0
template <typename R, typename SH, typename SY>
__device__ __forceinline__ R block_window_energy(const DevParams& P, const Sig<R>& G, SH& sh, int t, int& len, SY& sy)
{
    int s, e, es;
    len = centered_span(P.T, P.W, t, s, e, es);
    R p = (R)0, q = (R)0;
    if (len > 0) {
        const int n = len * P.F;
        const R* v = G.r + (int64_t)s * P.F;
        constexpr int kU = 8;                      // loads of a batch issued together
        for (int i0 = ltid(); i0 < n; i0 += kThreads * kU) {
            R x[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) { const int i = i0 + u * kThreads; x[u] = i < n ? v[i] : (R)0; }
#pragma unroll
            for (int u = 0; u < kU; ++u) { if (i0 + u * kThreads < n) { const R sq = x[u] * x[u]; p = p + sq; } }
        }
    }
    pinned_tree2(p, q, sh.red, sy);
    return p;
}

// The same energy by ONE wave, bit for bit: lane l carries the four strided partial sums l, 64+l, 128+l, 192+l
// (what threads l, 64+l, ... of the workgroup version carry), runs the per-wave halving tree on each and
// combines them as (P0+P1)+(P2+P3).  No workgroup barrier: the weak-atom filter gives every wave its own
// candidates.  Result valid in lane 0.
template <typename R>
__device__ __forceinline__ R wave_window_energy(const DevParams& P, const Sig<R>& G, int t, int& len, int lane)
{
    int s, e, es;
    len = centered_span(P.T, P.W, t, s, e, es);
    R p0 = (R)0, p1 = (R)0, p2 = (R)0, p3 = (R)0;
    if (len > 0) {
        const int n = len * P.F;
        const R* v = G.r + (int64_t)s * P.F;
        for (int i0 = lane; i0 < n; i0 += kThreads) {          // element i belongs to partial i mod 256
            const R x0 = v[i0];
            const R x1 = i0 + 64 < n ? v[i0 + 64] : (R)0;
            const R x2 = i0 + 128 < n ? v[i0 + 128] : (R)0;
            const R x3 = i0 + 192 < n ? v[i0 + 192] : (R)0;
            const R s0 = x0 * x0, s1 = x1 * x1, s2 = x2 * x2, s3 = x3 * x3;
            p0 = p0 + s0; p1 = p1 + s1; p2 = p2 + s2; p3 = p3 + s3;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const R o0 = __shfl_down(p0, m), o1 = __shfl_down(p1, m), o2 = __shfl_down(p2, m), o3 = __shfl_down(p3, m);
        p0 = p0 + o0; p1 = p1 + o1; p2 = p2 + o2; p3 = p3 + o3;
    }
    const R a01 = p0 + p1, a23 = p2 + p3;
    return a01 + a23;
}

// ------------------------------------------------------------------------------------------------
// GenericRecorr: re-correlate rows p-(W-1)..p+(W-1) against the reflect-padded residual span
// (modeling.py:1018-1051), reduce each row to its per-position best, write best_c/best_k.
// Threads = (row, atom group); the per-atom fma chain order is the pinned one (f outer, w inner).
// ------------------------------------------------------------------------------------------------
template <typename R> struct GenericRecorr {
    static constexpr int kMaxSegments = kMaxSeg;
    static constexpr bool kFused = false;               // uses the step-by-step atom body of iterate_kernel
    static constexpr bool kLocomp = false;              // (hscmp_locomp.h: LocompRecorr re-fits the atom's neighbourhood)
    static constexpr int kMinWavesPerSimd = 1;          // (register budget: no constraint)
    static constexpr int kEnergyWaves = kWaves;
    static constexpr bool kScoreOnly = false;           // keeps (coefficient, atom) per position
    using Shared = IterSharedT<R, kMaxSeg>;
    struct Args {};                                     // no extra kernel arguments
    static constexpr int kGroup = 1;                    // signals per workgroup
    using Sync = HwSync;
    static __device__ __forceinline__ Sync make_sync(Shared&) { return Sync(); }
    static __device__ __forceinline__ void prologue_shared(const DevParams&, const State<R>&, const Args&, char*) {}
    static __device__ __forceinline__ int signal_lds_offset(const DevParams&, const Args&) { return 0; }
    static __device__ __forceinline__ void epilogue(const DevParams&, const State<R>&, const Args&, char*, int) {}
    static __device__ __forceinline__ void resolve_wave(const DevParams&, const State<R>&, const Sig<R>&, const Args&, char*,
                                                        int, int, int&, R&) {}
    static __device__ __forceinline__ const R* weights(const DevParams&, const State<R>& S, const Args&, char*) { return S.weights; }
    static __device__ __forceinline__ void on_atom(const DevParams&, const State<R>&, const Args&, char*, int, int) {}
    static __device__ __forceinline__ bool update_residual(const DevParams&, const State<R>&, const Sig<R>&, const Args&, char*, int, int, R,
                                                           int, int, int, R&, R&) { return false; }
    static __device__ __forceinline__ bool window_partials(const DevParams&, const Sig<R>&, const Args&, char*, int, int, R&) { return false; }
    static __device__ __forceinline__ bool wave_window_listed(const DevParams&, const Sig<R>&, const Args&, char*, int, int, int, int, R&) { return false; }
    static __device__ __forceinline__ bool row_results(const DevParams&, const Args&, char*, int, const int*&, const R*&, const R*&, int&, int&) { return false; }
    static __device__ __forceinline__ bool residual_copy_in_lds(const Args&, char*) { return false; }
    static constexpr int kWinBytes = 16384;            // LDS window of the residual span, when it fits
    static size_t extra_lds_bytes(const DevParams&) { return kWinBytes; }
    static __device__ __forceinline__ void prologue(const DevParams&, const State<R>&, const Args&, char*, int, Sync&) {}
    template <typename SH>
    static __device__ __forceinline__ void run(const DevParams& P, const State<R>& S, const Sig<R>& G,
                                               SH& sh, const Args&, char* lds, int p, int)
    {
        const int T = P.T, K = P.K, W = P.W, F = P.F, tid = threadIdx.x;
        const int nrows = 2 * W - 1;
        const int tstart = p - P.off - (W - 1);            // :1028-1033
        const int tend = p + W / 2 + (W - 1);              // :1038
        const int sidx = tstart < 0 ? 0 : tstart;          // :1034
        const int eidx = tend > T - 1 ? T - 1 : tend;      // :1039
        const int nslice = eidx - sidx + 1;
        const bool interior = (tstart >= 0) && (tend <= T - 1);
        // reflect-padded residual span (np.pad 'reflect', :1046) staged in LDS when it fits
        R* win = reinterpret_cast<R*>(lds);
        const int span = 3 * W - 2;
        const bool staged = (size_t)span * F * sizeof(R) <= (size_t)kWinBytes;
        if (staged) {
            for (int i = tid; i < span * F; i += kThreads) {
                const int jj = i / F, ff = i - jj * F;
                const int gi = interior ? tstart + jj : reflect_index(tstart + jj, sidx, nslice);
                win[ff * span + jj] = G.r[(int64_t)gi * F + ff];     // feature-major: lanes (consecutive rows) read consecutive words
            }
            __syncthreads();
        }
        const int rpp = nrows < kThreads ? nrows : kThreads;
        int ngrp = kThreads / rpp;
        if (ngrp > K) ngrp = K;
        if (ngrp < 1) ngrp = 1;
        const R* __restrict__ D = S.D;
        const R* __restrict__ Dc = S.Dc;                    // [K][F][W]: consecutive along the chain
        for (int base = 0; base < nrows; base += rpp) {
            const int j = base + tid % rpp;
            const int g = tid / rpp;
            const int t = p - (W - 1) + j;
            const bool rowok = (j < nrows) && (t >= 0) && (t < T);   // overlapReplace clipping, utils.py:133-161
            R bs = (R)-1, bc = (R)0;
            int bk = 0;
            if (rowok && g < ngrp) {
                const int k0 = (int)(((int64_t)K * g) / ngrp), k1 = (int)(((int64_t)K * (g + 1)) / ngrp);
                int k = k0;
                for (; k < k1; ++k) {
                    const R* dk = D + (int64_t)k * W * F;
                    R acc = (R)0;
                    if (staged) {
                        const R* wj = win + j;                       // row j's window starts at span row j
                        const R* dc = Dc + (int64_t)k * W * F;
                        for (int f = 0; f < F; ++f) {
#pragma unroll 8
                            for (int w = 0; w < W; ++w) acc = rfma(wj[f * span + w], dc[f * W + w], acc);
                        }
                    } else {
                        for (int f = 0; f < F; ++f)
                            for (int w = 0; w < W; ++w) {
                                int gi = t - P.off + w;
                                if (!interior) gi = reflect_index(gi, sidx, nslice);   // np.pad 'reflect', :1046
                                acc = rfma(G.r[(int64_t)gi * F + f], dk[w * F + f], acc);
                            }
                    }
                    const R sc = score_of(acc, k, S.weights);
                    if (sc > bs) { bs = sc; bc = acc; bk = k; }
                }
            }
            sh.part_s[tid] = bs; sh.part_c[tid] = bc; sh.part_k[tid] = bk;
            __syncthreads();
            if (g == 0 && rowok) {
                for (int q = 1; q < ngrp; ++q) {       // ascending atom groups, strict > keeps the lowest k
                    const int o = q * rpp + tid;
                    if (sh.part_s[o] > bs) { bs = sh.part_s[o]; bc = sh.part_c[o]; bk = sh.part_k[o]; }
                }
                G.bc[t] = bc;
                G.bk[t] = bk;
            }
            __syncthreads();
        }
    }
};

// ------------------------------------------------------------------------------------------------
// the greedy loop            grid = ceil(B / kGroup), block = kGroup * kThreads
//   256 threads (4 waves) work on one signal from its first selection to its last.  kGroup = 1: the signal owns
//   the workgroup.  kGroup = 4 (hscmp_mfma.h): four signals share a 1024-thread workgroup and ONE dictionary
//   image in LDS; each signal synchronises its own four waves (SoftSync), the signals never wait for each other.
// ------------------------------------------------------------------------------------------------
// (hscmp_locomp.h: the atom body of LoCOMP, modeling.py:1314-1383, and what is computed ahead for the selections of a blocked round
//  that lie far enough apart: lists and fitted coefficients of a group, lane i of the owning wave holding atom i)
template <typename R> struct LocompPre {
    int status, n, t, k, si; R a;       // status 2: lists + fitted coefficients ready; 3: also applied on a private copy of the residual:
    int u0, ulen; R loss; R span[4];    //   samples [u0, u0 + ulen) (lane l holds l, l + 64, ...) and the group's energy loss
    int cell[2];                        //   (sparse dictionary instead: up to two cells per lane, their final values in span[0..1]; -1: none)
};
constexpr int kLocompSpacing = 5;      // x W + 8 samples between any two selections of a round that are computed side by side
// the rows of a selection whose re-correlation waits for the end of its batch (the owning wave's registers)
struct LocompRows { int pending, pmin, pmax; };
template <typename R, typename Pol, typename SH, typename SY>
__device__ __forceinline__ int locomp_atom(const DevParams& P, const State<R>& S, const Sig<R>& G, SH& sh, const typename Pol::Args& A,
                                           char* plds, const R* wts, int p, int k, R c, SY& sy, const LocompPre<R>& pre, int owner,
                                           bool may_defer, LocompRows& rows);      // bit 0: the re-correlation of its rows waits; bit 2: committed by its wave
                                                                                   // alone, and bit 1 says whether the pursuit of this signal ends with it
template <typename R, typename Pol, typename SY>
__device__ __forceinline__ void locomp_rows_deferred(const DevParams& P, const State<R>& S, const Sig<R>& G, const typename Pol::Args& A, char* plds,
                                                     LocompRows& rows, SY& sy);
template <typename R, typename Pol, typename SH, typename SY>
__device__ __forceinline__ void locomp_precompute(const DevParams& P, const State<R>& S, const Sig<R>& G, SH& sh, const typename Pol::Args& A, char* plds,
                                                  const int* ord_t, const int* ord_k, const R* ord_c, int first, int count, LocompPre<R>& pre,
                                                  LocompRows& rows, SY& sy);

template <typename R, typename Recorr>
__global__ __launch_bounds__(kThreads * Recorr::kGroup, Recorr::kMinWavesPerSimd) void iterate_kernel(DevParams P, State<R> S, typename Recorr::Args A)
{
    // all LDS comes from ONE dynamic array (16-byte aligned base): per signal the control block first, then the
    // policy's region (dictionary image, residual window); a policy with kGroup > 1 keeps what the signals share
    // in front of the per-signal regions
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using SH = typename Recorr::Shared;
    constexpr int GS = Recorr::kGroup;
    Recorr::prologue_shared(P, S, A, smem);              // (kGroup > 1: the shared image, behind a hardware barrier)
    // (wave-uniform by construction; readfirstlane tells the compiler, so the per-signal pointers live in SGPRs)
    const int b = GS == 1 ? (int)blockIdx.x : (int)blockIdx.x * GS + gsig();
    if (GS > 1 && b >= P.B) return;                      // (a ragged last workgroup; no hardware barrier from here on)
    char* sbase = smem + Recorr::signal_lds_offset(P, A);
    SH& sh = *reinterpret_cast<SH*>(sbase);
    char* plds = sbase + ((sizeof(SH) + 15) / 16) * 16;
    int tid = ltid(), lane = tid & 63, wv = tid >> 6;     // (re-derived at the top of every round: see laundered_tid)
    typename Recorr::Sync sy = Recorr::make_sync(sh);
    int* stats = S.stats + (int64_t)b * ST_COUNT;
    if (stats[ST_STOP] != STOP_RUNNING) return;          // converged in an earlier launch

    const int T = P.T, W = P.W, F = P.F;
    Sig<R> G;
    G.r = S.residual + (int64_t)b * T * F;
    G.bc = S.best_c + (int64_t)b * T;
    G.bk = S.best_k + (int64_t)b * T;
    G.ev_t = S.ev_t + (int64_t)b * P.cap; G.ev_k = S.ev_k + (int64_t)b * P.cap; G.ev_c = S.ev_c + (int64_t)b * P.cap;
    G.slot_t = S.slot_t + (int64_t)b * P.cap; G.slot_k = S.slot_k + (int64_t)b * P.cap; G.slot_a = S.slot_a + (int64_t)b * P.cap;
    G.hkey = S.hkey + (int64_t)b * ((int64_t)P.hmask + 1); G.hval = S.hval + (int64_t)b * ((int64_t)P.hmask + 1);
    G.sel_t = S.sel_t + (int64_t)b * 2 * P.maxsel; G.sel_k = S.sel_k + (int64_t)b * 2 * P.maxsel; G.sel_c = S.sel_c + (int64_t)b * 2 * P.maxsel;
    G.head = (Recorr::kLocomp && S.head) ? S.head + (int64_t)b * T : S.head;
    G.lgram = S.lgram ? S.lgram + (int64_t)b * (int64_t)lgram_doubles(P.lg_cap) : nullptr;
#ifdef HSCMP_DBG_STAMPS
    if (tid == 0 && b < 4096) {
        g_blk[3 * b + 0] = wall_clock64();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_blk[3 * b + 2] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    Recorr::prologue(P, S, A, plds, b, sy);
    const R* wts = Recorr::weights(P, S, A, plds);        // atom weights (a policy may keep a copy in LDS)

    // ---- prologue: segment maxima of the per-position best (and of |residual|)
    for (int sg = wv; sg < P.nseg; sg += kWaves) scan_segment<Recorr::kScoreOnly>(P, G, wts, sh, sg, lane);
    if (P.has_scale) for (int sg = wv; sg < P.nseg; sg += kWaves) rscan_segment(P, G, sh, sg, lane);
    if (tid == 0) {
        sh.nnz = stats[ST_NNZ]; sh.ndup = stats[ST_DUP]; sh.rounds = stats[ST_ROUNDS]; sh.iters = stats[ST_ITERS];
        sh.nev = stats[ST_EVENTS]; sh.nslots = stats[ST_SLOTS]; sh.offset = stats[ST_OFFSET];
        sh.converged = 0; sh.stop = STOP_RUNNING; sh.nsel = 0; sh.skip = 0; sh.found = -1; sh.nullsel = 0; sh.hashed = 0; sh.fpos = 0;
        sh.e_sig = S.energy[2 * b + 0]; sh.e_res = S.energy[2 * b + 1];
        sh.ctl[0] = 0; sh.ctl[1] = sh.nev; sh.ctl[2] = sh.nslots; sh.ctl[3] = 0;
    }
    if constexpr (Recorr::kLocomp) {
        // LoCOMP: the coefficient slots chained by position (head[t] -> most recent slot, hval[slot] -> the one before): its
        // neighbourhood search walks them (hscmp_locomp.h).  Rebuilt on every launch from the slot list, like the Bloom filter below.
        for (int i = tid; i < T; i += kThreads) hval_store(G.head + i, -1);
        sy.full();
        const int ns = stats[ST_SLOTS];
        for (int i = tid; i < ns; i += kThreads)
            hval_store(G.hval + i, __hip_atomic_exchange(G.head + G.slot_t[i], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if constexpr (!Recorr::kFused) {
        for (int i = tid; i < (Recorr::kMaxSegments + 31) / 32; i += kThreads) sh.touched[i] = 0u;
        // Bloom filter over the (t,k) pairs that already own a coefficient slot (rebuilt on every launch)
        for (int i = tid; i < kBloomWords; i += kThreads) sh.bloom[i] = 0u;
        sy.full();
        const int ns = stats[ST_SLOTS];
        for (int i = tid; i < ns; i += kThreads) {
            const unsigned h = bloom_hash(G.slot_t[i], G.slot_k[i]);
            atomicOr(&sh.bloom[h >> 5], 1u << (h & 31));
        }
    }
    sy.full();

    // candidate lists of a round: LDS when they fit (the compactions and sorts of the blocked selection are a chain
    // of dependent passes over them), else the per-signal global buffers; the select-only entry point hands the
    // ordered list back through the global buffer
    const bool sel_lds = P.maxsel <= kSelLds && !P.select_only;
    int* raw_t = sel_lds ? sh.sel_t : G.sel_t; int* raw_k = sel_lds ? sh.sel_k : G.sel_k; R* raw_c = sel_lds ? sh.sel_c : G.sel_c;   // first half
    int* ord_t = raw_t + P.maxsel; int* ord_k = raw_k + P.maxsel; R* ord_c = raw_c + P.maxsel;                                      // second half
    const double thres = P.thres;
    const bool has_thres = P.has_thres != 0;

    // several signals per workgroup: a signal's serial code outranks the other signals' tiles (hscmp_mfma.h, apply_atom)
    if constexpr (GS > 1) __builtin_amdgcn_s_setprio(3);

    FusedCtl fc = fused_ctl_fetch(sh);                   // (behind the barrier above; only the fused bodies keep it current)
    HSCMP_STAMP_BEGIN();
    for (int round = 0; P.max_rounds <= 0 || round < P.max_rounds; ++round) {
        if constexpr (Recorr::kMinWavesPerSimd >= 4) asm volatile("" : "+v"(tid));     // (register-constrained builds only)
        lane = tid & 63; wv = tid >> 6;
        int nsel;
        if constexpr (!Recorr::kFused) HSCMP_STAMP(39); else HSCMP_STAMP(10);     // fused: from the atom's return to the next round
        // =========================== select (modeling.py:899-982) ===========================
        int p_sel = 0, k_sel = 0;
        R c_sel = (R)0;
        if (!P.blocked) {
            HSCMP_MARK("select");
            // :965-975 flat arg-max == arg-max over the segment maxima (ties: lowest t, then k)
            if constexpr (Recorr::kFused) {
                // EVERY wave scans the segment maxima for itself: no barrier, nothing to publish.  (A barrier of this kernel
                // is an LDS atomic plus at least one polling read, and beside the matrix instructions of the co-resident
                // signals every LDS round trip of a wave takes ~1000 cycles -- tools/serial_stretch_probe.hip; the scan is
                // one batch of reads and an arg-max.)  Lane l looks at the `per` consecutive segments from l * per on -- lanes
                // in index order, see wave_argmax_first -- and carries the position of its best.
                Cand<R> c; c.s = (R)-1; c.i = INT_MAX;
                const int per = (P.nseg + 63) >> 6;
                for (int j0 = 0; j0 < per; j0 += 4) {                 // ascending i per lane: '>' keeps the first of equals
                    // four maxima and their positions in ONE batch of LDS reads (clamped index, unconditional: a guarded
                    // read per element compiles to a chain of dependent round trips)
                    R sv[4]; int tv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = min(lane * per + j0 + u, P.nseg - 1);
                        sv[u] = sh.seg_score[i]; tv[u] = sh.seg_t[i];
                    }
                    asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (j0 + u < per && lane * per + j0 + u < P.nseg && sv[u] > c.s) { c.s = sv[u]; c.i = tv[u]; }
                }
                c = wave_argmax_first(c);
                p_sel = c.i;
                static_assert(Recorr::kScoreOnly, "the fused atom body resolves (k, c) itself");
                // (k, c) and the null test (:974) are resolved inside apply_atom.  No position at all: every score is NaN
                // (a pursuit that diverged until the residual overflowed) -- nothing is selected and the loop ends
                nsel = p_sel == INT_MAX ? 0 : 1;
            } else {
                Cand<R> c; c.s = (R)-1; c.i = INT_MAX;
                for (int i = tid; i < P.nseg; i += kThreads) {
                    Cand<R> o; o.s = sh.seg_score[i]; o.i = i;
                    if (better(o, c)) c = o;
                }
                c = wave_argmax(c);
                if (lane == 0) sh.cred[wv] = c;
                sy.full();
                if (tid == 0) {
                    Cand<R> m = sh.cred[0];
                    for (int q = 1; q < kWaves; ++q) if (better(sh.cred[q], m)) m = sh.cred[q];
                    const int sg = m.i == INT_MAX ? 0 : m.i;          // (no segment compares: NaN scores of a diverged pursuit)
                    const R cc = sh.seg_c[sg];          // (non-fused policies keep coefficient / atom per segment)
                    sh.atom_t = sh.seg_t[sg]; sh.atom_k = sh.seg_k[sg]; sh.atom_c = cc;
                    sh.nsel = (m.i == INT_MAX || (has_thres && !(fabs((double)cc) > thres))) ? 0 : 1;     // :974
                }
                sy.full();
                nsel = sh.nsel;
                p_sel = sh.atom_t; k_sel = sh.atom_k; c_sel = sh.atom_c;
            }
        } else {
            // :908-937 one arg-max per block of bs samples (half-block shifted when offset is set)
#ifdef HSCMP_DBG_CHECKSEG
            // diagnostic build: the segment maxima kept in LDS against a fresh scan of the score array
            if constexpr (Recorr::kFused) {
                for (int sg = wv; sg < P.nseg; sg += kWaves) {
                    const int t0 = sg << P.seg_shift, t1 = min(T, t0 + P.seg);
                    Cand<R> w = wave_range_argmax<Recorr::kScoreOnly>(G, wts, t0, t1, lane);
                    if (w.i == INT_MAX) { w.i = t0; w.s = (R)0; }
                    if (lane == 0 && (w.s != sh.seg_score[sg] || w.i != sh.seg_t[sg])) {
                        const unsigned long long n = atomicAdd(&g_cnt[0], 1ull);
                        if (n < 3) {
                            g_cnt[4 + 4 * n + 0] = ((unsigned long long)b << 32) | (unsigned)sg;
                            g_cnt[4 + 4 * n + 1] = ((unsigned long long)__float_as_uint((float)w.s) << 32) | __float_as_uint((float)sh.seg_score[sg]);
                            g_cnt[4 + 4 * n + 2] = ((unsigned long long)(unsigned)w.i << 32) | (unsigned)sh.seg_t[sg];
                            g_cnt[4 + 4 * n + 3] = ((unsigned long long)(unsigned)sh.rounds << 32) | (unsigned)sh.iters;
                        }
                    }
                }
                sy.lds();
            }
#endif
            const int off = sh.offset;
            const int nb = P.nbk + (off ? 1 : 0);
            const int pad0 = off ? P.bs / 2 : 0;
            for (int j = wv; j < nb; j += kWaves) {
                const int w0 = j * P.bs - pad0;
                const int lo = w0 < 0 ? 0 : w0;
                const int hi = min(T, w0 + P.bs);
                Cand<R> win; win.s = (R)-1; win.i = INT_MAX;
                if (lo < hi) win = wave_block_argmax<Recorr::kScoreOnly>(P, G, wts, sh, lo, hi, lane);
                bool valid = (lo < hi) && win.i != INT_MAX;                      // :940-942 range test
                if (valid && win.s == (R)0 && w0 < 0) valid = false;             // arg-max on a leading padded row
                int wk = 0;
                R wc = (R)0;
                if (valid) {                                                     // wave-uniform
                    if constexpr (Recorr::kScoreOnly) Recorr::resolve_wave(P, S, G, A, plds, win.i, lane, wk, wc);
                    else { wk = G.bk[win.i]; wc = G.bc[win.i]; }
                }
                if (lane == 0) { raw_t[j] = valid ? win.i : -1; raw_k[j] = wk; raw_c[j] = wc; }
            }
            sy.full();
            if constexpr (!Recorr::kFused) HSCMP_STAMP(48);
            // :946-948 drop null coefficients (and invalid blocks): raw -> ord
            int n = block_compact(nb, [&](int i) { return raw_t[i] >= 0 && (!has_thres || fabs((double)raw_c[i]) > thres); },
                                  raw_t, raw_k, raw_c, ord_t, ord_k, ord_c, sh, sy);
            // :951-957 interference filter vs the unfiltered predecessor; skipped when no gap qualifies
            if (n > 1) {
                int cnt = 0;
                for (int i = 1 + tid; i < n; i += kThreads) cnt += (ord_t[i] - ord_t[i - 1] >= W) ? 1 : 0;
                cnt = sy.count(cnt > 0);   // > 0 iff some gap qualifies
                if (cnt > 0) {
                    n = block_compact(n, [&](int i) { return i == 0 || (ord_t[i] - ord_t[i - 1] >= W); },
                                      ord_t, ord_k, ord_c, raw_t, raw_k, raw_c, sh, sy);
                } else {
                    n = block_compact(n, [&](int) { return true; }, ord_t, ord_k, ord_c, raw_t, raw_k, raw_c, sh, sy);
                }
            } else {
                n = block_compact(n, [&](int) { return true; }, ord_t, ord_k, ord_c, raw_t, raw_k, raw_c, sh, sy);
            }
            // :960-962 argsort(|c|)[::-1]: descending, the later entry first among equals (rank sort)
            for (int i = tid; i < n; i += kThreads) {
                const R a = rabs(raw_c[i]);
                int rank = 0;
                for (int q = 0; q < n; ++q) {
                    const R o = rabs(raw_c[q]);
                    rank += (o > a || (o == a && q > i)) ? 1 : 0;
                }
                ord_t[rank] = raw_t[i]; ord_k[rank] = raw_k[i]; ord_c[rank] = raw_c[i];
            }
            sy.full();
            if constexpr (!Recorr::kFused) HSCMP_STAMP(49);
            // :1090-1099 weak-atom filter
            if (P.has_snr && n > 1) {
                const R tol_energy = sh.e_sig / (R)P.snr_ratio;
                const double thr = (double)tol_energy / (double)((int64_t)T * F);
                if (W * F <= 4 * kThreads) {
                    // short windows: one wave per candidate, no workgroup barriers (same sums, see wave_window_energy)
                    for (int i = wv; i < n; i += kWaves) {
                        int len;
                        const R e = wave_window_energy(P, G, ord_t[i], len, lane);
                        if (lane == 0) {
                            const R mean = e / (R)((int64_t)len * F);
                            raw_t[i] = ((double)mean >= thr) ? 1 : 0;      // keep flag (raw half is free now)
                        }
                    }
                } else {
                    // long windows (multi-feature inputs).  A policy that knows which cells can be non-zero gives every
                    // wave its own candidates first (flag -1: not settled that way) ...
                    // (a policy whose LDS table holds fewer than four windows lets the waves in by turns)
                    constexpr int kEW = Recorr::kEnergyWaves;
                    for (int i0 = 0; i0 < n; i0 += kWaves) {
                        for (int turn = 0; turn < kWaves / kEW; ++turn) {
                            const int i = i0 + wv;
                            if (wv / kEW == turn && i < n) {
                                int len, ws, we, wes;
                                len = centered_span(T, W, ord_t[i], ws, we, wes);
                                R e = (R)0;
                                const bool done = len > 0 && Recorr::wave_window_listed(P, G, A, plds, ws, we, lane, wv % kEW, e);
                                if (lane == 0) {
                                    const R mean = e / (R)((int64_t)len * F);
                                    raw_t[i] = !done ? -1 : ((double)mean >= thr) ? 1 : 0;
                                }
                            }
                            if (kEW < kWaves) sy.lds();
                        }
                    }
                    sy.full();
                    // ... and the whole workgroup streams what is left, window by window
                    for (int i = 0; i < n; ++i) {
                        if (raw_t[i] >= 0) continue;                       // uniform
                        int len, ws, we, wes;
                        len = centered_span(T, W, ord_t[i], ws, we, wes);
                        R e = (R)0, q2 = (R)0;
                        if (len > 0 && Recorr::window_partials(P, G, A, plds, ws, we, e)) pinned_tree2(e, q2, sh.red, sy);   // listed cells only
                        else e = block_window_energy(P, G, sh, ord_t[i], len, sy);
                        if (tid == 0) {
                            const R mean = e / (R)((int64_t)len * F);
                            raw_t[i] = ((double)mean >= thr) ? 1 : 0;
                        }
                    }
                }
                sy.full();
                // keep flags live in raw_t; entry o <= i is only overwritten after flag i was read
                // (block_compact evaluates a whole 256-chunk of predicates before it writes)
                n = block_compact(n, [&](int i) { return raw_t[i] != 0; }, ord_t, ord_k, ord_c, raw_t, raw_k, raw_c, sh, sy);
                n = block_compact(n, [&](int) { return true; }, raw_t, raw_k, raw_c, ord_t, ord_k, ord_c, sh, sy);
            }
            if (tid == 0) sh.nsel = n;
            sy.full();
            nsel = sh.nsel;
        }

        HSCMP_MARK("after_select");
        if (P.select_only) {
            // _selectBestAtoms entry point (hscmp_select_best_atoms): the ordered list goes back as is
            if (tid == 0) {
                if (!P.blocked && nsel > 0) { ord_t[0] = p_sel; ord_k[0] = k_sel; ord_c[0] = c_sel; }
                stats[ST_EVENTS] = nsel;
            }
            return;
        }
        if constexpr (!Recorr::kFused) HSCMP_STAMP(32);
        // A round whose atoms do not all fit the event list is not started: the state then is exactly that
        // of a round boundary, and hscmp_grow_events + hscmp_continue resume bit for bit.
        const int nev_now = Recorr::kFused ? fc.nev : sh.nev, nslots_now = Recorr::kFused ? fc.nslots : sh.nslots;
        const bool hashed_now = Recorr::kFused ? fc.hashed != 0 : sh.hashed != 0;
        const bool lists_full = nev_now + nsel > P.cap;  // uniform (LDS values after a barrier)
        // the slot list has outgrown the Bloom filter: duplicate lookups go through the hash table from here on
        const bool build_table = !hashed_now && nslots_now >= P.hash_min;
        const bool hashed = hashed_now || build_table;
        // every thread has read the event / slot counts before thread 0 advances them (the step-by-step body may
        // reach its bookkeeping without passing another barrier; the fused bodies pass several first)
        if constexpr (!Recorr::kFused) sy.full();
        if (lists_full) {
            sy.full();
            if (tid == 0) { sh.converged = 1; sh.stop = STOP_CAPACITY; }
            break;
        }
        if (build_table) {
            sy.full();                         // (drains the deferred slot stores of the fused bodies)
            slot_table_build(G, P.hmask, sh.nslots, sy);
            if (tid == 0) { sh.hashed = 1; sh.ctl[3] = 1; }
            fc.hashed = 1;
            sy.full();
        }
        if constexpr (!Recorr::kFused) {
            // The pairs of one round are distinct (one per block), so the table as of the round start answers for all
            // of them: one lookup per atom, side by side.  raw_t (free after the selection) carries the answers.
            if (hashed && nsel > 0) {
                if (!P.blocked) {
                    if (tid == 0) { unsigned pos; sh.found = slot_find(G, P.hmask, p_sel, k_sel, pos); sh.fpos = pos; }
                } else {
                    for (int i = tid; i < nsel; i += kThreads) { unsigned pos; raw_t[i] = slot_find(G, P.hmask, ord_t[i], ord_k[i], pos); }
                    sy.full();
                }
            }
        }
        // =========================== apply the selected atoms (:1101-1142) ===========================
        bool fused_stop = false;       // uniform: apply_atom's return value is read after its last barrier
        // LoCOMP: are the selections of this round more than 5W + 8 samples apart, every pair of them?  (hscmp_locomp.h)
        bool lc_spaced = false;
        int lc_first = 0, lc_count = 0;
        LocompPre<R> lc_pre{};
        LocompRows lc_rows{0, 0, 0};
        bool lc_defer = false, lc_waiting = false;       // uniform: this policy's state allows deferred rows; some wave holds rows that wait
        if constexpr (Recorr::kLocomp) {
            if (P.blocked && nsel >= 2 && !P.select_only && (P.lc_ahead & 1)) {
                int bad = 0;
                for (int e = tid; e < nsel * nsel; e += kThreads) {
                    const int i = e / nsel, j = e - i * nsel;
                    if (i < j && abs(ord_t[i] - ord_t[j]) <= kLocompSpacing * W + 8) bad = 1;
                }
                lc_spaced = sy.count(bad) == 0;
                lc_defer = lc_spaced && (P.lc_ahead & 4) && Recorr::can_defer_rows(P, A, plds);
            }
        }
        for (int ai = 0; ai < nsel; ++ai) {
            int p, k; R c;
            if (!P.blocked) { p = p_sel; k = k_sel; c = c_sel; }
            else { p = ord_t[ai]; k = ord_k[ai]; c = ord_c[ai]; }
            if constexpr (Recorr::kFused) {
                // policy-owned atom body: one batch of global loads, LDS-only barriers (hscmp_mfma.h);
                // in blocked mode (k, c) were resolved at selection time
                HSCMP_STAMP(11);                                   // selection + the round's checks
                const bool stop_now = Recorr::apply_atom(P, S, G, sh, A, plds, p, k, c, P.blocked != 0, sy, fc);
                HSCMP_STAMP(12);                                   // the atom body, entry checks included
                if (stop_now) { fused_stop = true; break; }
                continue;
            }
            if constexpr (Recorr::kLocomp) {
                // selections far enough apart: neighbourhoods, normal equations and re-fits of up to kWaves of them at once, one
                // wave each (locomp_precompute), then the applications in order
                if (lc_spaced && ai == lc_first + lc_count && nsel - ai >= 2) {
                    lc_first = ai; lc_count = min(kWaves, nsel - ai);
                    locomp_precompute<R, Recorr>(P, S, G, sh, A, plds, ord_t, ord_k, ord_c, lc_first, lc_count, lc_pre, lc_rows, sy);
                    lc_waiting = false;                  // (rows of the previous batch that waited were its waves' first job)
                }
                const bool ahead = lc_spaced && ai >= lc_first && ai < lc_first + lc_count;
                // the rows of the selections of a batch are re-correlated TOGETHER behind its last one, one wave per selection: nothing a
                // later selection of the batch reads or writes lies within the rows of an earlier one, or within the samples they are formed from
                const int lc_rc = locomp_atom<R, Recorr>(P, S, G, sh, A, plds, wts, p, k, c, sy, lc_pre, ahead ? ai - lc_first : -1, ahead && lc_defer, lc_rows);
                if (lc_rc & 1) lc_waiting = true;
                // (uniform: read behind the atom's last barrier -- or handed over by it when no barrier separates it from the next selection's writes)
                const bool leave = (lc_rc & 4) ? (lc_rc & 2) != 0 : (sh.skip || sh.converged);
                // (when another batch of this round follows, its computations ahead take the rows along: neither reads what the other writes,
                //  and a wave with many rows to do may well own a small group next)
                if (lc_waiting && (leave || ai == nsel - 1 || (ai == lc_first + lc_count - 1 && nsel - (ai + 1) < 2))) {
                    locomp_rows_deferred<R, Recorr>(P, S, G, A, plds, lc_rows, sy);
                    lc_waiting = false;
                }
                if (leave) break;
                continue;
            }

            // ---- :1106-1114 duplicate / nnz bookkeeping, coefficient accumulation, event append
            // the slot list is searched only when the Bloom filter says the pair may own a slot already
            const unsigned hb = bloom_hash(p, k);
            const bool maybe_dup = !hashed && ((sh.bloom[hb >> 5] >> (hb & 31)) & 1u) != 0;    // uniform (LDS, ordered by the barriers below)
            if (maybe_dup) {
                if (tid == 0) sh.found = -1;
                sy.full();
                const int nslots = sh.nslots;
                for (int i = tid; i < nslots; i += kThreads)
                    if (G.slot_t[i] == p && G.slot_k[i] == k) sh.found = i;          // at most one match
                sy.full();
            }
            bool new_slot = false;                   // thread 0 only
            if (tid == 0) {
                if (sh.nev >= P.cap) { sh.converged = 1; sh.stop = STOP_CAPACITY; sh.skip = 1; }
                else {
                    int si = hashed ? (P.blocked ? raw_t[ai] : sh.found) : (maybe_dup ? sh.found : -1);
                    if (si >= 0 && fabs(G.slot_a[si]) > 0.0) sh.ndup += 1;
                    else if (rabs(c) > (R)0) sh.nnz += 1;
                    if (si < 0) {
                        si = sh.nslots++; G.slot_t[si] = p; G.slot_k[si] = k; new_slot = true;
                        if (hashed) {
                            // one atom per round: the probe's free entry is still free.  Blocked rounds enter their
                            // new slots together at the round end (raw_t <= -2 names the slot).
                            if (!P.blocked) slot_insert_at(G, sh.fpos, p, k, si); else raw_t[ai] = -2 - si;
                        }
                        G.slot_a[si] = 0.0 + (double)c;                 // (:992 starts the accumulator at 0.0)
                    } else {
                        G.slot_a[si] += (double)c;
                    }
                    const int e = sh.nev++;
                    G.ev_t[e] = p; G.ev_k[e] = k; G.ev_c[e] = c;
                }
            }
            // LDS-only barrier: nobody waits for thread 0's list stores here (the duplicate search that reads the
            // slot list comes after later full barriers), only for the control block
            sy.lds();
            // the filter is updated only now: every thread has read this atom's bit (maybe_dup) before the barrier
            // above, and the next read comes after the barriers of the residual update
            if (new_slot && !hashed) sh.bloom[hb >> 5] |= 1u << (hb & 31);
            if (sh.skip) break;
            if constexpr (!Recorr::kFused) HSCMP_STAMP(33);

            // (policy hook: work that only needs to know the atom, overlapped with the residual update below)
            Recorr::on_atom(P, S, A, plds, p, k);

            // ---- :1117, :996-1016 residual subtract with local energy before / after
            int s, e, es;
            const int len = centered_span(T, W, p, s, e, es);
            R pb = (R)0, pa = (R)0;
            // (a policy that knows which cells can be non-zero updates only those and returns its energy partials)
            if (!Recorr::update_residual(P, S, G, A, plds, p, k, c, s, e, es, pb, pa)) {
                const int n = len * F;
                const R nc = -c;
                const R* dk = S.D + ((int64_t)k * W + es) * F;
                R* rv = G.r + (int64_t)s * F;
                // (the loads of a batch are issued together: one memory round trip per 8 elements of a thread)
                constexpr int kU = 8;
                for (int i0 = tid; i0 < n; i0 += kThreads * kU) {
                    R v[kU], d[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int i = i0 + u * kThreads;
                        v[u] = (R)0; d[u] = (R)0;
                        if (i < n) { v[u] = rv[i]; d[u] = dk[i]; }
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int i = i0 + u * kThreads;
                        if (i < n) {
                            const R sq = v[u] * v[u];
                            pb = pb + sq;
                            const R prod = nc * d[u];        // -c*D[k] rounded, then += (utils.py:120,129)
                            const R vn = v[u] + prod;
                            rv[i] = vn;
                            const R sq2 = vn * vn;
                            pa = pa + sq2;
                        }
                    }
                }
            }
            pinned_tree2(pb, pa, sh.red, sy);
            if (tid == 0) {
                const R loss = pb - pa;              // :1005
                sh.e_res = sh.e_res - loss;          // :1014
            }
            // residual writes visible to the whole workgroup -- unless the policy re-correlates from its own LDS copy of
            // the window and nothing else reads the residual before the atom's last barrier
            if (!P.has_scale && Recorr::residual_copy_in_lds(A, plds)) sy.lds(); else sy.full();
            if (P.has_scale) {
                const int sg0 = s >> P.seg_shift, sg1 = (e - 1) >> P.seg_shift;
                for (int sg = sg0 + wv; sg <= sg1; sg += kWaves) rscan_segment(P, G, sh, sg, lane);
            }

            if constexpr (!Recorr::kFused) HSCMP_STAMP(34);
            // ---- :1120, :1018-1051 local re-correlation of the 2W-1 touched rows
            Recorr::run(P, S, G, sh, A, plds, p, k);
            // (a policy that still holds the rows' results in LDS lets the segment scan start before their stores land)
            const int* rows_k = nullptr; const R* rows_c = nullptr; const R* rows_0 = nullptr;
            int rows_t0 = 0, rows_n = 0;
            bool rows_lds = false;
            if constexpr (!Recorr::kFused && !Recorr::kScoreOnly) rows_lds = Recorr::row_results(P, A, plds, p, rows_k, rows_c, rows_0, rows_t0, rows_n);
            if (!rows_lds) sy.full();
            if constexpr (!Recorr::kFused) HSCMP_STAMP(35);

            // ---- refresh the maxima of the touched segments
            {
                const int lo = max(0, p - (W - 1)), hi = min(T - 1, p + (W - 1));
                const int sg0 = lo >> P.seg_shift, sg1 = hi >> P.seg_shift;
                if (P.blocked) {
                    // the maxima are only read by the next selection: the atoms of a blocked round mark their segments,
                    // which are rescanned once at the round end (no global round trip per atom)
                    if (tid == 0) for (int sg = sg0; sg <= sg1; ++sg) sh.touched[sg >> 5] |= 1u << (sg & 31);
                } else if (rows_lds) {
                    if constexpr (!Recorr::kScoreOnly)
                        for (int sg = sg0 + wv; sg <= sg1; sg += kWaves) scan_segment_rows(P, G, wts, sh, sg, lane, rows_k, rows_c, rows_0, rows_t0, rows_n);
                } else {
                    for (int sg = sg0 + wv; sg <= sg1; sg += kWaves) scan_segment<Recorr::kScoreOnly>(P, G, wts, sh, sg, lane);
                }
            }

            // ---- :1122-1142 fast stop rules
            if (tid == 0) {
                sh.iters += 1;
                if ((double)sh.e_res < P.eps) { sh.converged = 1; sh.stop = STOP_ENERGY_EPS; }
                else if (P.l0 >= 0 && sh.nnz >= P.l0) { sh.converged = 1; sh.stop = STOP_NNZ; }
                else if (P.has_snr) {
                    const R q = sh.e_sig / sh.e_res;
                    if ((double)q >= P.snr_ratio) { sh.converged = 1; sh.stop = STOP_SNR; }
                }
            }
            sy.full();
            if constexpr (!Recorr::kFused) { HSCMP_STAMP(36); if (b == 0 && tid == 0) HSCMP_COUNT(46); }
            if (sh.converged) break;
        }

        if constexpr (!Recorr::kFused) {
            // blocked rounds: maxima of the segments the round's atoms touched (their stores are behind the atoms' barriers)
            if (P.blocked && nsel > 0 && !sh.converged) {
                // every wave walks the set bits (lane l holds word l) and takes the segments sg = wave (mod 4)
                constexpr int kWords = (Recorr::kMaxSegments + 31) / 32;
                static_assert(kWords <= 64, "one mask word per lane");
                const unsigned word = lane < kWords ? sh.touched[lane] : 0u;
                unsigned long long words = __ballot(word != 0u);
                while (words) {                                             // wave-uniform
                    const int wl = __ffsll((long long)words) - 1;
                    words &= words - 1ull;
                    unsigned bits = __shfl(word, wl);
                    while (bits) {
                        const int sg = wl * 32 + __ffs((int)bits) - 1;
                        bits &= bits - 1u;
                        if ((sg & (kWaves - 1)) != wv) continue;
                        scan_segment<Recorr::kScoreOnly>(P, G, wts, sh, sg, lane);
                        if (lane == 0) atomicAnd(&sh.touched[sg >> 5], ~(1u << (sg & 31)));
                    }
                }
            }
            // blocked rounds: the new slots of the round enter the table (thread 0's notes are behind the atoms' barriers)
            if (hashed && P.blocked)
                for (int i = tid; i < nsel; i += kThreads) {
                    const int v = raw_t[i];
                    if (v <= -2) slot_insert(G, P.hmask, ord_t[i], ord_k[i], -2 - v);
                }
        }
        // =========================== slow stop rules (:1145-1163) ===========================
        if constexpr (Recorr::kFused) {
            // single arg-max rounds without a residual-scale rule need no further synchronisation:
            // the stop decision already went through apply_atom's last barrier
            if (!P.blocked && !P.has_scale && nsel > 0) {
                // (the round counter and the offset toggle of such a round are advanced by the atom body's bookkeeping)
                if (fused_stop) break;
                continue;
            }
        }
        if (P.has_scale) {
            R m = (R)0;
            for (int i = tid; i < P.nseg; i += kThreads) { const R a = sh.rseg[i]; m = a > m ? a : m; }
            m = wave_max(m);
            sy.full();
            if (lane == 0) sh.red[wv] = m;
            sy.full();
            if (tid == 0) {
                for (int q = 1; q < kWaves; ++q) m = sh.red[q] > m ? sh.red[q] : m;
                if ((double)m <= P.tol_scale) { sh.converged = 1; if (sh.stop == STOP_RUNNING) sh.stop = STOP_SCALE; }
            }
        }
        if (tid == 0) {
            if (nsel == 0 || sh.nullsel) { sh.converged = 1; if (sh.stop == STOP_RUNNING) sh.stop = STOP_EMPTY; }   // (after the scale rule, :1145-1153)
            sh.rounds += 1;
            sh.offset = !sh.offset;
        }
        sy.full();
        if (sh.converged) break;
    }

    sy.full();
#ifdef HSCMP_DBG_STAMPS
    if (tid == 0 && b < 4096) g_blk[3 * b + 1] = wall_clock64();
#endif
    Recorr::epilogue(P, S, A, plds, b);
    if (tid == 0) {
        stats[ST_NNZ] = sh.nnz; stats[ST_DUP] = sh.ndup; stats[ST_ROUNDS] = sh.rounds; stats[ST_STOP] = sh.stop;
        stats[ST_ITERS] = sh.iters; stats[ST_EVENTS] = sh.nev; stats[ST_SLOTS] = sh.nslots; stats[ST_OFFSET] = sh.offset;
        S.energy[2 * b + 1] = sh.e_res;
    }
}

}  // namespace hscmp
