// hscmp_epilogue.h -- epilogue of the hierarchical encoder on the device.
//
// After the last level of HierarchicalConvolutionalMatchingPursuit (hsc/modeling.py:1427-1654) the reference
//   * turns the level's accumulated coefficients into a CSC matrix, dropping zeros and |c| < minCoefficients (:1171-1181),
//   * hands the singleton columns of that matrix back to the levels they pass through (convertToDistributedCoefficients,
//     :1556-1594: level l owns a contiguous column range of the LAST level's matrix, same column numbers),
//   * synthesises every level through its input-level representations and subtracts the sum from the input
//     (_calculateResidual :1596-1611, reconstructSignal :226-263, overlapAdd utils.py:103-131),
//   * and, for storage, lists the coefficients as (time, level, index, value) records sorted by time (dataset.py:798-811).
// All of that is a function of ONE slot list per signal (the last level's distinct (t, column) pairs with their float64
// sums), so one workgroup per signal does it where the list lives:
//   1. filter + key (column, t, slot) -> bitonic sort            = CSC order (column-major, rows ascending)
//   2. indices / data / column pointers out
//   3. key (t, column, CSC rank) -> bitonic sort                 = the reference's event order (time, level, index)
//   4. event records out
//   5. per sample: the entries whose pattern covers it, level by level, each level in CSC order -- the order in which the
//      reference's sequential overlap-add reaches that sample -- float64 sums: bit-identical to hscmp_host_overlap_add.
//      Optionally the energy of the residual (sum of squares, fixed order) instead of / next to its samples: a caller that
//      only checks the reconstruction quality need not move T samples per signal over PCIe.
// Keys are 64-bit with the payload in the low bits (column 20 | t 24 | index 20), sorted in LDS when the list fits
// (<= 16384 entries) and in a global scratch row otherwise.
#pragma once

#include "hscmp_device.h"

namespace hscmp {

constexpr int kEpiThreads = 1024;
constexpr int kEpiMaxLevels = 8;
constexpr int kEpiLdsKeys = 16384;            // 128 KB of keys
constexpr int kEpiColBits = 20, kEpiTBits = 24, kEpiIdxBits = 20;

struct EpiLevel {
    int col0, col1;       // columns [col0, col1) of the last level's matrix belong to this level (col1 <= col0: none)
    int scale, lead;      // taps of the level's input-level patterns; (scale - 1) / 2 (utils.py:84-99)
    const void* rep;      // device: [K][scale][Fd]
    int rep_f32;
    int pad_;
};

struct EpiArgs {
    EpiLevel lv[kEpiMaxLevels];
    int nlevels, Ktot, T, Fd;
    int has_min;
    double minc;
    int max_back, max_fwd;               // an entry at t can cover samples t - max_back' ... : window of t around a sample s is [s - max_back, s + max_fwd]
    // the last level's slots
    const int* slot_t; const int* slot_k; const double* slot_a; const int* stats; int cap;
    // outputs (device)
    const long long* offsets;            // [count + 1] entry offsets of the packed arrays
    int* out_n;                          // [count]
    int* out_colptr;                     // [count][Ktot + 1]
    int* out_indices; double* out_data;  // packed, CSC order
    int* out_events;                     // packed 16-byte records (t, level, index, float value), or nullptr
    double* out_residual;                // [count][T][Fd] or nullptr
    double* out_energy;                  // [count] sum of the squared residual samples, or nullptr
    unsigned long long* scratch;         // [count][scratch_n] keys of lists that do not fit LDS
    int scratch_n;
    int lds_keys;                        // keys that fit the dynamic LDS of the launch (a power of two)
    int lds_bytes;                       // that LDS in bytes (the residual tiles of step 5 reuse it)
};

__device__ __forceinline__ int epi_level_of(const EpiArgs& A, int col)
{
    int l = 0;
    for (int q = 0; q < A.nlevels; ++q) if (col >= A.lv[q].col0 && col < A.lv[q].col1) l = q;
    return l;
}

// in-place ascending bitonic sort of N (power of two) 64-bit keys by the whole workgroup
__device__ __forceinline__ void epi_sort(unsigned long long* keys, int N)
{
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = threadIdx.x; p < (N >> 1); p += kEpiThreads) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int q = i | j;
                const unsigned long long a = keys[i], b = keys[q];
                const bool asc = (i & k) == 0;
                if ((a > b) == asc) { keys[i] = b; keys[q] = a; }
            }
            __syncthreads();
        }
    }
}

// the same sort for a list that lives in global memory and is longer than the LDS holds (`cap` keys, a power of two): every
// run of compare-exchange steps with a distance below `cap` is carried out chunk by chunk in LDS -- load, all the steps,
// store -- and only the steps with a larger distance pass over global memory (N = 2 cap: one such pass in all).
__device__ __forceinline__ void epi_sort_chunked(unsigned long long* keys, int N, unsigned long long* lds, int cap)
{
    // stages k <= cap: every chunk is sorted on its own, in the direction its position in the final network asks for
    for (int c0 = 0; c0 < N; c0 += cap) {
        for (int p = threadIdx.x; p < cap; p += kEpiThreads) lds[p] = keys[c0 + p];
        __syncthreads();
        for (int k = 2; k <= cap; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int p = threadIdx.x; p < (cap >> 1); p += kEpiThreads) {
                    const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    const int q = i | j;
                    const unsigned long long a = lds[i], b = lds[q];
                    const bool asc = ((c0 + i) & k) == 0;
                    if ((a > b) == asc) { lds[i] = b; lds[q] = a; }
                }
                __syncthreads();
            }
        for (int p = threadIdx.x; p < cap; p += kEpiThreads) keys[c0 + p] = lds[p];
        __syncthreads();
    }
    for (int k = cap << 1; k <= N; k <<= 1) {
        for (int j = k >> 1; j >= cap; j >>= 1) {                // distances that cross chunks: over global memory
            for (int p = threadIdx.x; p < (N >> 1); p += kEpiThreads) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int q = i | j;
                const unsigned long long a = keys[i], b = keys[q];
                const bool asc = (i & k) == 0;
                if ((a > b) == asc) { keys[i] = b; keys[q] = a; }
            }
            __syncthreads();
        }
        for (int c0 = 0; c0 < N; c0 += cap) {                    // the rest of the merge inside each chunk
            for (int p = threadIdx.x; p < cap; p += kEpiThreads) lds[p] = keys[c0 + p];
            __syncthreads();
            for (int j = cap >> 1; j > 0; j >>= 1) {
                for (int p = threadIdx.x; p < (cap >> 1); p += kEpiThreads) {
                    const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    const int q = i | j;
                    const unsigned long long a = lds[i], b = lds[q];
                    const bool asc = ((c0 + i) & k) == 0;
                    if ((a > b) == asc) { lds[i] = b; lds[q] = a; }
                }
                __syncthreads();
            }
            for (int p = threadIdx.x; p < cap; p += kEpiThreads) keys[c0 + p] = lds[p];
            __syncthreads();
        }
    }
}

template <typename XR>
__global__ __launch_bounds__(kEpiThreads) void hier_epilogue_kernel(EpiArgs A, const XR* __restrict__ x)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_kept;
    __shared__ double s_energy[kEpiThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nslots = A.stats[(int64_t)b * ST_COUNT + ST_SLOTS];
    int N = 2;
    while (N < nslots) N <<= 1;
    // (A.lds_keys: how many keys the LDS of this launch holds -- kEpiLdsKeys, less under HSCMP_EPI_LDS_KEYS in the tests)
    unsigned long long* lds_keys = reinterpret_cast<unsigned long long*>(smem);
    const bool in_lds = N <= A.lds_keys;
    unsigned long long* keys = in_lds ? lds_keys : A.scratch + (int64_t)b * A.scratch_n;
    const int* st = A.slot_t + (int64_t)b * A.cap;
    const int* sk = A.slot_k + (int64_t)b * A.cap;
    const double* sa = A.slot_a + (int64_t)b * A.cap;
    const long long off = A.offsets[b];
    if (tid == 0) s_kept = 0;
    __syncthreads();

    // ---- 1. filter (:1171-1181) and sort by (column, t)
    int kept = 0;
    for (int i = tid; i < N; i += kEpiThreads) {
        unsigned long long key = ~0ull;
        if (i < nslots) {
            const double a = sa[i];
            if (a != 0.0 && !(A.has_min && !(fabs(a) >= A.minc))) {
                key = ((unsigned long long)(unsigned)sk[i] << (kEpiTBits + kEpiIdxBits)) | ((unsigned long long)(unsigned)st[i] << kEpiIdxBits) | (unsigned)i;
                ++kept;
            }
        }
        keys[i] = key;
    }
    if (kept) atomicAdd(&s_kept, kept);
    __syncthreads();
    const int nk = s_kept;
    if (in_lds) epi_sort(keys, N); else epi_sort_chunked(keys, N, lds_keys, A.lds_keys);

    // ---- 2. CSC out: indices, data, column pointers over the last level's columns
    int* colptr = A.out_colptr + (int64_t)b * (A.Ktot + 1);
    for (int j = tid; j <= nk; j += kEpiThreads) {
        const int cprev = j > 0 ? (int)(keys[j - 1] >> (kEpiTBits + kEpiIdxBits)) : -1;
        const int c = j < nk ? (int)(keys[j] >> (kEpiTBits + kEpiIdxBits)) : A.Ktot;
        for (int q = cprev + 1; q <= c; ++q) colptr[q] = j;
        if (j < nk) {
            const int idx = (int)(keys[j] & ((1u << kEpiIdxBits) - 1u));
            A.out_indices[off + j] = (int)((keys[j] >> kEpiIdxBits) & ((1u << kEpiTBits) - 1u));
            A.out_data[off + j] = sa[idx];
        }
    }
    if (tid == 0) A.out_n[b] = nk;
    __syncthreads();                                   // (also: out_data is visible to the whole workgroup)

    // ---- 3. re-key by (t, column) with the CSC rank as payload; sort
    for (int j = tid; j < N; j += kEpiThreads) {
        unsigned long long key = ~0ull;
        if (j < nk) {
            const unsigned long long k1 = keys[j];
            const unsigned long long col = k1 >> (kEpiTBits + kEpiIdxBits), t = (k1 >> kEpiIdxBits) & ((1u << kEpiTBits) - 1u);
            key = (t << (kEpiColBits + kEpiIdxBits)) | (col << kEpiIdxBits) | (unsigned)j;
        }
        // (every thread rewrites only the entries it read: same index j)
        keys[j] = key;
    }
    __syncthreads();
    if (in_lds) epi_sort(keys, N); else epi_sort_chunked(keys, N, lds_keys, A.lds_keys);

    // ---- 4. event records in the reference's order: time, then level, then index (dataset.py:798-811)
    if (A.out_events) {
        for (int i = tid; i < nk; i += kEpiThreads) {
            const unsigned long long k3 = keys[i];
            const int t = (int)(k3 >> (kEpiColBits + kEpiIdxBits));
            const int col = (int)((k3 >> kEpiIdxBits) & ((1u << kEpiColBits) - 1u));
            const int j = (int)(k3 & ((1u << kEpiIdxBits) - 1u));
            int* ev = A.out_events + 4 * (off + i);
            ev[0] = t; ev[1] = epi_level_of(A, col); ev[2] = col;
            ev[3] = __float_as_int((float)A.out_data[off + j]);
        }
    }

    // ---- 5. residual: x - sum over the levels of their synthesis, each level's terms in CSC order
    if (A.out_residual || A.out_energy) {
        const int64_t nel = (int64_t)A.T * A.Fd;
        const XR* xb = x + (int64_t)b * nel;
        double* rb = A.out_residual ? A.out_residual + (int64_t)b * nel : nullptr;
        const double* data = A.out_data + off;
        double esum = 0.0;                                   // this thread's samples, in index order
        // One-dimensional signals: tile by tile in LDS.  The entries whose patterns reach a tile of TS samples are a range of
        // the t-sorted keys; sorted by CSC rank they fall into one run per level (a level owns a range of columns).  Every
        // wave owns a strip of the tile and walks the run of the level IN THAT ORDER, adding the taps that land on its strip
        // (the order in which the reference's sequential overlap-add reaches each sample); then recon += level, level after
        // level, and the thread that owns sample s (s mod 1024, as in the per-sample form below: the energy sum keeps its
        // order) takes x - recon.  A tile with more entries than the LDS list holds takes the per-sample form.
        int TS = 0;
        if (A.Fd == 1 && nk > 0) {
            TS = 4096;
            while (TS > 256 && (size_t)TS * 32 > (size_t)A.lds_bytes) TS >>= 1;       // sig + recon + (key, value) per entry
            if ((size_t)TS * 32 > (size_t)A.lds_bytes || A.max_back + A.max_fwd + TS >= (1 << 13)) TS = 0;
        }
        const unsigned long long* gk = keys;
        if (TS > 0 && in_lds) {
            // the keys move to the global scratch row: the LDS is about to hold the tiles
            unsigned long long* row = A.scratch + (int64_t)b * A.scratch_n;
            for (int i = tid; i < nk; i += kEpiThreads) row[i] = keys[i];
            gk = row;
            __syncthreads();
        }
        auto per_sample = [&](int64_t e) {
            const int s = (int)(e / A.Fd), fd = (int)(e - (int64_t)s * A.Fd);
            // first entry with t >= s - max_back (binary search over the t-sorted keys)
            const long long tlo = (long long)s - A.max_back;
            int lo = 0, hi = nk;
            if (tlo > 0) {
                const unsigned long long bound = (unsigned long long)tlo << (kEpiColBits + kEpiIdxBits);
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (gk[mid] < bound) lo = mid + 1; else hi = mid; }
            }
            const int i0 = lo;
            const long long thi = (long long)s + A.max_fwd;
            double recon = 0.0;
            for (int l = 0; l < A.nlevels; ++l) {
                const EpiLevel& L = A.lv[l];
                if (L.col1 <= L.col0) continue;
                double sig = 0.0;
                int last = -1;
                for (;;) {
                    int best = INT_MAX, bt = 0, bcol = 0;
                    for (int i = i0; i < nk; ++i) {
                        const unsigned long long k3 = gk[i];
                        const long long t = (long long)(k3 >> (kEpiColBits + kEpiIdxBits));
                        if (t > thi) break;
                        const int col = (int)((k3 >> kEpiIdxBits) & ((1u << kEpiColBits) - 1u));
                        const int j = (int)(k3 & ((1u << kEpiIdxBits) - 1u));
                        if (col < L.col0 || col >= L.col1 || j <= last || j >= best) continue;
                        const long long w = (long long)s - (t - L.lead);          // tap of the pattern that lands on s
                        if (w < 0 || w >= L.scale) continue;
                        best = j; bt = (int)t; bcol = col;
                    }
                    if (best == INT_MAX) break;
                    const int w = s - (bt - L.lead);
                    const int64_t ri = ((int64_t)bcol * L.scale + w) * A.Fd + fd;
                    const double rv = L.rep_f32 ? (double)reinterpret_cast<const float*>(L.rep)[ri] : reinterpret_cast<const double*>(L.rep)[ri];
                    const double prod = data[best] * rv;                          // c * D[k] rounded, then += (utils.py:120,129)
                    sig = sig + prod;
                    last = best;
                }
                recon = recon + sig;                                             // reconstruction += reconstructSignal(level), :1606-1608
            }
            const double rv = (double)xb[e] - recon;
            if (rb) rb[e] = rv;
            esum = esum + rv * rv;
        };
        if (TS == 0) {
            for (int64_t e = tid; e < nel; e += kEpiThreads) per_sample(e);
        } else {
            double* sig = reinterpret_cast<double*>(smem);                        // [TS]
            double* recon = sig + TS;                                             // [TS]
            unsigned long long* ekey = reinterpret_cast<unsigned long long*>(recon + TS);   // [TS]  CSC rank | t - tbase | column
            double* eval = reinterpret_cast<double*>(ekey + TS);                  // [TS]
            __shared__ int s_range[2 + kEpiMaxLevels + 1];
            const int lane = tid & 63, wv = tid >> 6, strip = TS / (kEpiThreads / 64);
            for (int t0 = 0; t0 < A.T; t0 += TS) {
                const int tn = min(TS, A.T - t0);
                const long long tbase = (long long)t0 - A.max_back;
                if (tid < 2) {
                    // entries with t in [t0 - max_back, t0 + tn - 1 + max_fwd]
                    const long long tq = tid == 0 ? tbase : (long long)t0 + tn + A.max_fwd;
                    int lo = 0, hi = nk;
                    if (tq > 0) {
                        const unsigned long long bound = (unsigned long long)tq << (kEpiColBits + kEpiIdxBits);
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (gk[mid] < bound) lo = mid + 1; else hi = mid; }
                    }
                    s_range[tid] = lo;
                }
                __syncthreads();
                const int i0 = s_range[0], ne = s_range[1] - s_range[0];
                if (ne > TS) {                                                    // (uniform) too dense for the list: per sample
                    for (int s = t0 + tid; s < t0 + tn; s += kEpiThreads) per_sample(s);
                    __syncthreads();
                    continue;
                }
                int NE = 2;
                while (NE < ne) NE <<= 1;
                for (int i = tid; i < NE; i += kEpiThreads) {
                    unsigned long long key = ~0ull;
                    if (i < ne) {
                        const unsigned long long k3 = gk[i0 + i];
                        const long long t = (long long)(k3 >> (kEpiColBits + kEpiIdxBits));
                        const unsigned long long col = (k3 >> kEpiIdxBits) & ((1u << kEpiColBits) - 1u), j = k3 & ((1u << kEpiIdxBits) - 1u);
                        key = (j << (13 + kEpiColBits)) | ((unsigned long long)(t - tbase) << kEpiColBits) | col;
                    }
                    ekey[i] = key;
                }
                for (int s = tid; s < TS; s += kEpiThreads) recon[s] = 0.0;
                __syncthreads();
                epi_sort(ekey, NE);
                for (int i = tid; i < ne; i += kEpiThreads) eval[i] = data[(int)(ekey[i] >> (13 + kEpiColBits))];
                // run of every level in the rank-sorted list (ranks ascend with the column)
                if (tid <= A.nlevels) {
                    int pos = ne;
                    if (tid < A.nlevels) {
                        const int c0 = A.lv[tid].col0;
                        int lo = 0, hi = ne;
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)(ekey[mid] & ((1u << kEpiColBits) - 1u)) < c0) lo = mid + 1; else hi = mid; }
                        pos = lo;
                    }
                    s_range[2 + tid] = pos;
                }
                __syncthreads();
                for (int l = 0; l < A.nlevels; ++l) {
                    const EpiLevel& L = A.lv[l];
                    if (L.col1 <= L.col0) continue;
                    for (int s = tid; s < TS; s += kEpiThreads) sig[s] = 0.0;
                    __syncthreads();
                    // this level's run: from its first column to the first entry of a column beyond its last
                    const int r0 = s_range[2 + l];
                    const int sub0 = t0 + wv * strip, sub1 = min(t0 + tn, sub0 + strip);
                    for (int i = r0; i < ne; ++i) {                               // (uniform per wave)
                        const unsigned long long key = ekey[i];
                        const int col = (int)(key & ((1u << kEpiColBits) - 1u));
                        if (col >= L.col1) break;
                        const int t = (int)(tbase + (long long)((key >> kEpiColBits) & 0x1fffu));
                        const int first = t - L.lead;
                        const int lo = max(first, sub0), hi = min(first + L.scale, sub1);
                        if (lo >= hi) continue;
                        const double c = eval[i];
                        for (int s = lo + lane; s < hi; s += 64) {
                            const int64_t ri = (int64_t)col * L.scale + (s - first);
                            const double rv = L.rep_f32 ? (double)reinterpret_cast<const float*>(L.rep)[ri] : reinterpret_cast<const double*>(L.rep)[ri];
                            const double prod = c * rv;                           // c * D[k] rounded, then += (utils.py:120,129)
                            sig[s - t0] = sig[s - t0] + prod;
                        }
                    }
                    __syncthreads();
                    for (int s = tid; s < tn; s += kEpiThreads) recon[s] = recon[s] + sig[s];      // reconstruction += reconstructSignal(level)
                    __syncthreads();
                }
                for (int s = t0 + tid; s < t0 + tn; s += kEpiThreads) {
                    const double rv = (double)xb[s] - recon[s - t0];
                    if (rb) rb[s] = rv;
                    esum = esum + rv * rv;
                }
                __syncthreads();
            }
        }
        if (A.out_energy) {
            // fixed summation order: per thread over its strided samples, xor tree inside the wave, the 16 wave sums in order
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) esum = esum + __shfl_xor(esum, m);
            if ((tid & 63) == 0) s_energy[tid >> 6] = esum;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
                for (int w = 0; w < kEpiThreads / 64; ++w) tot = tot + s_energy[w];
                A.out_energy[b] = tot;
            }
        }
    }
}

}  // namespace hscmp
