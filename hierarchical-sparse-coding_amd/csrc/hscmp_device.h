// hscmp_device.h -- device-side building blocks shared by every kernel variant of the engine.
//
// Table-free formulation: the reference keeps the whole inner-product table
// ip[T,K] (hsc/modeling.py:1077) and re-scans it for every selection (:967).  The engine keeps,
// per position t, only the entry that can win the arg-max: (best_c[t], best_k[t]) = the
// coefficient and atom index of max_k |ip[t,k]*w_k| (first k on ties), plus per-segment maxima
// of those.  Every local update replaces whole rows of ip (:1049), so recomputing the 2W-1
// touched rows and their per-position best is observably identical to the reference.
#pragma once

#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <stdint.h>
#include <limits.h>

namespace hscmp {

#ifdef HSCMP_DBG_CHECKSEG
#define HSCMP_DBG_STAMPS
#endif
#ifdef HSCMP_DBG_STAMPS
// diagnostic build only (tools/read_stamps.py); never compiled into the product library
__device__ unsigned long long g_stamps[64];      // per-phase cycle sums of workgroup 0: [0,16) fused atom body, [32,48) step-by-step body and sparse_rows
__device__ unsigned long long g_blk[3 * 4096];   // per workgroup: start, end (100 MHz wall clock), XCC/HW id
__device__ unsigned long long g_cnt[16];         // free-form event counters of workgroup 0
#endif
#ifdef HSCMP_DBG_STAMPS
// diagnostic build only: per-phase cycle sums of workgroup 0 (thread 0), read back by
// tools/read_stamps.py / tools/hsc_stamps.py through hscmp_debug_stamps(); never compiled into the product library
#define HSCMP_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_stamps[i] += now_ - stamp_last_; stamp_last_ = now_; } } while (0)
#define HSCMP_STAMP_BEGIN() unsigned long long stamp_last_ = clock64()
#define HSCMP_COUNT(i) (g_stamps[i] += 1)
#define HSCMP_TALLY(i, v) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_cnt[i] += (unsigned long long)(v); } while (0)
#else
#define HSCMP_STAMP(i) do {} while (0)
#define HSCMP_STAMP_BEGIN() do {} while (0)
#define HSCMP_COUNT(i) ((void)0)
#define HSCMP_TALLY(i, v) do {} while (0)
#endif


// analysis build only (-DHSCMP_MARKS): comment lines in the assembly that delimit the source phases, so that
// tools/count_marks.py can attribute instruction counts; emits no instruction
#ifdef HSCMP_MARKS
#define HSCMP_MARK(name) asm volatile("; HSCMP_MARK " name ::: "memory")
#else
#define HSCMP_MARK(name) do {} while (0)
#endif

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the
// outstanding global stores (vmcnt), which costs a full memory round trip per barrier
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// Synchronisation of the threads that work on ONE signal.
//   HwSync    the signal owns the whole 256-thread workgroup: the hardware barrier.
//   SoftSync  several signals share one workgroup (hscmp_mfma.h, four signals per 1024-thread workgroup: one
//             dictionary image per CU, four waves per SIMD): s_barrier would put the signals in lockstep, so the four
//             waves of a signal meet at a monotone counter in LDS instead.  One lane per wave adds 1; the wave then
//             polls until the counter reaches 4 x (barriers passed).  The LDS executes the operations of a wave in
//             issue order, so everything a wave wrote (or read) before its add is done before a wave that has seen the
//             add issues its next access: no s_waitcnt is needed for LDS data.  full() also drains the wave's global
//             stores / loads (s_waitcnt vmcnt(0)) before the add, as __syncthreads() does: the waves of a workgroup
//             share the CU's write-through vector cache.
// Every wave of the group must pass the same sequence of barriers.
// ------------------------------------------------------------------------------------------------
// index of a thread inside the 256 threads that work on its signal, and the signal of a thread inside its workgroup.
// 256-thread kernels: threadIdx.x itself.  Four signals per 1024-thread workgroup: signal i owns threads 256i..256i+255
// -- waves 4i..4i+3, which the hardware spreads over the four SIMDs (tools/simd_map_probe.hip: waves w, w+4, w+8, w+12
// of a workgroup share a SIMD) -- and its waves are rotated by i roles, so that the wave that does a signal's one-lane
// bookkeeping (and every other per-wave role) sits on a different SIMD for each of the four signals.
// -DHSCMP_QUAD_SIMD_AFFINE=1 (measurement only) gives every signal one SIMD to itself instead (signal = wave & 3): its
// serial phases then never meet another signal's matrix instructions, but its four waves also share one vector ALU
// and nothing fills the matrix pipe of that SIMD meanwhile -- measured 10.7 ms against 9.6 ms for the greedy loop of
// config 2.
#ifndef HSCMP_QUAD_SIMD_AFFINE
#define HSCMP_QUAD_SIMD_AFFINE 0
#endif
__device__ __forceinline__ int ltid()
{
#if HSCMP_QUAD_SIMD_AFFINE
    return blockDim.x > 256u ? (int)(((threadIdx.x >> 8) << 6) | (threadIdx.x & 63u)) : (int)threadIdx.x;
#else
    return (int)((threadIdx.x + ((threadIdx.x >> 8) << 6)) & 255u);
#endif
}
__device__ __forceinline__ int gsig()
{
#if HSCMP_QUAD_SIMD_AFFINE
    return __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) & 3u));
#else
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
#endif
}

// threadIdx.x the compiler cannot see through: what is derived from it inside a per-atom function is recomputed there
// instead of being hoisted out of the atom loop (where it would be live across everything and, in the 128-VGPR builds,
// spilled -- every reload a memory round trip on a latency-bound path)
// (ON only in the builds that are short of registers: where there is room the hoisted values cost nothing and the
// recomputation does -- measured 2 % on the roomy level loop)
template <bool ON> __device__ __forceinline__ int laundered_tid()
{
    int t = (int)threadIdx.x;
    if constexpr (ON) asm volatile("" : "+v"(t));
    return t;
}

struct HwSync {
    static constexpr int kGroup = 1;
    __device__ __forceinline__ void lds() { lds_barrier(); }
    __device__ __forceinline__ void full() { __syncthreads(); }
    __device__ __forceinline__ int count(int pred) { return __syncthreads_count(pred); }
};

struct SoftSync {
    static constexpr int kGroup = 4;
    unsigned addr;        // LDS byte address of the counter (monotone, compared modulo 2^32); wave-uniform
    int* cnt;             // LDS: scratch of count()
    unsigned target;      // counter value once all four waves have arrived at this wave's current barrier (wave-uniform)
    // Hand-written: the vector ALU is what the other signals' f32 MFMA tiles compete for, and the compiler's form of
    // "one lane adds, all lanes poll" costs 7 + 3 x polls vector instructions per barrier.  Here the arrival is scalar
    // code around one ds_add (EXEC narrowed to lane 0; the increment is the counter's own address -- any non-zero
    // constant does, and that one is already in a register), and a poll is ds_read + v_readfirstlane.
    __device__ __forceinline__ void init(unsigned* bar, int* cnt_)
    {
        addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)bar);
        cnt = cnt_;
        target = 0u;
    }
    __device__ __forceinline__ void arrive_and_wait()
    {
        target = __builtin_amdgcn_readfirstlane(target) + 4u * addr;
        unsigned long long saved;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %1\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(addr) : "memory");
        for (int nap = 0;; ++nap) {               // look at once, then at growing intervals (64 .. 512 cycles)
            unsigned v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            if ((int)(__builtin_amdgcn_readfirstlane(v) - target) >= 0) break;
            if (nap < 2) __builtin_amdgcn_s_sleep(1);
            else if (nap < 4) __builtin_amdgcn_s_sleep(2);
            else if (nap < 8) __builtin_amdgcn_s_sleep(4);
            else __builtin_amdgcn_s_sleep(8);
        }
    }
    __device__ __forceinline__ void lds() { arrive_and_wait(); }
    __device__ __forceinline__ void full()
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        arrive_and_wait();
    }
    // number of threads of the group whose predicate holds (as __syncthreads_count)
    __device__ __forceinline__ int count(int pred)
    {
        const int n = __popcll(__ballot(pred != 0));
        if (ltid() == 0) *cnt = 0;
        full();
        if ((threadIdx.x & 63) == 0 && n) atomicAdd(cnt, n);
        lds();
        const int total = *cnt;
        lds();                  // everybody has read the total before the next count() resets it
        return total;
    }
};

constexpr int kEdgeWords = 4;    // per signal: left mask, right mask, stale sample index + 1 (0: none), its saved bits
constexpr int kThreads = 256;   // one workgroup = 4 waves of 64
constexpr int kWaves = kThreads / 64;
constexpr int kMaxSeg = 1024;   // segment maxima kept in LDS per signal

// Kernel parameters (by value).  Per-signal arrays are [B][...] with the strides below.
struct DevParams {
    int B, T, K, W, F;
    int off;            // (W-1)/2: lead of the centred window (modeling.py:159-164, utils.py:84-99)
    int seg, nseg;      // segment size (power of two >= 64) and count, nseg <= kMaxSeg
    int seg_shift;      // log2(seg)
    // selection (modeling.py:899-982)
    int blocked;        // 0: single arg-max; 1: blocked
    int bs, nbk;        // block size (even) and base block count ceil(T/bs)
    int maxsel;         // capacity of the per-round selection list
    // stop rules (modeling.py:1125-1158)
    int l0;             // nbNonzeroCoefs or -1
    int has_snr, has_scale, has_thres;
    double snr_ratio;   // 10^(toleranceSnr/10): snr >= tol <=> Esig/Eres >= snr_ratio
    double tol_scale;
    double thres;       // nullCoeffThres
    double eps;
    int cap;            // max events per signal
    int hash_min;       // slot count from which the loop keeps that table (kSlotHashMin; HSCMP_SLOT_HASH_MIN overrides)
    unsigned hmask;     // slots of the per-signal (t,k) -> coefficient-slot hash table, minus one (power of two >= 2*cap)
    int max_rounds;     // <= 0: until converged
    int select_only;    // 1: run ONE selection (modeling.py:899-982), hand the atoms back, apply nothing
    int lg_cap;         // LoCOMP: atoms of the largest group the signal's global scratch holds (lgram_doubles; larger: STOP_GROUP)
    int lc_ahead;       // LoCOMP: bit 2 = the rows of a batch of selections re-correlated behind its last one; bit 0 = the selections of a round that lie far enough apart are computed side by side, bit 1 = a group's rows
                        // re-correlated one wave per quarter on sparse dictionaries (hscmp_locomp.h; HSCMP_LOCOMP_AHEAD)
};

// LoCOMP: doubles of global scratch per signal for a group of up to `cap` atoms (hscmp_locomp.h, GroupGlobal): two packed triangles
// (Gram matrix / factor, S of the minimum-norm completion), four double vectors, six int lists
constexpr int kLocompGroupCap = 512;
__host__ __device__ inline size_t lgram_doubles(int cap) { return (size_t)cap * ((size_t)cap + 1) + 4 * (size_t)cap + 3 * (size_t)cap; }

// table size of the slot hash: load factor <= 1/2 whatever the event list holds
inline unsigned slot_hash_mask(int cap)
{
    unsigned h = 64;
    while (h < 2u * (unsigned)cap) h <<= 1;
    return h - 1;
}

// segment size: the smallest power of two >= 64 that keeps the segment count within maxseg
inline void set_segments(DevParams& P, int maxseg)
{
    int seg = 64, shift = 6;
    while ((P.T + seg - 1) / seg > maxseg) { seg <<= 1; ++shift; }
    P.seg = seg;
    P.seg_shift = shift;
    P.nseg = (P.T + seg - 1) / seg;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) raises a ceiling: once a kernel has been allowed `lds` bytes on a device
// every smaller request is covered.  The call takes the host some tens of microseconds and used to sit between two
// queued kernels of every encode (a gap on the GPU whenever the host is not ahead of it), as did the occupancy query of
// the persistent initial correlation: both are answered from a cache after the first time.
inline hipError_t set_dyn_lds(const void* kern, size_t lds)
{
    // process-wide and monotone: hipFuncSetAttribute SETS the ceiling of a kernel for every thread of the process, so a
    // thread asking for less than another thread was granted must not lower it (several host threads drive engines of
    // their own: the LoCOMP workers of the hierarchical batch entry point)
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> allowed;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = allowed[std::make_pair(dev, kern)];
    if (have >= lds && have > 0) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) have = lds;
    return e;
}
inline int cached_blocks_per_cu(const void* kern, int threads, size_t lds)
{
    static thread_local std::map<std::tuple<int, const void*, int, size_t>, int> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    auto key = std::make_tuple(dev, kern, threads, lds);
    auto it = seen.find(key);
    if (it != seen.end()) return it->second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    seen[key] = per_cu;
    return per_cu;
}

enum { ST_NNZ = 0, ST_DUP = 1, ST_ROUNDS = 2, ST_STOP = 3, ST_ITERS = 4, ST_EVENTS = 5, ST_SLOTS = 6,
       ST_OFFSET = 7, ST_COUNT = 8 };
enum { STOP_RUNNING = 0, STOP_ENERGY_EPS = 1, STOP_NNZ = 2, STOP_SNR = 3, STOP_SCALE = 4, STOP_EMPTY = 5,
       STOP_CALLBACK = 6, STOP_CAPACITY = 7,
       STOP_STALLED = 8,       // LoCOMP: an atom changed the residual energy by less than eps (modeling.py:1379-1383)
       STOP_GROUP = 9 };       // LoCOMP: a neighbourhood larger than the kernel re-fits (kLocompMax): the host loop takes the signal

template <typename R> struct State {
    const R* D;         // [K][W][F]
    const R* Dc;        // [K][F][W]: the same dictionary in chain order (f outer, w inner); == D when F == 1
    const R* weights;   // [K] or nullptr
    R* residual;        // [B][T*F]
    R* best_c;          // [B][T]
    int* best_k;        // [B][T]
    int* ev_t; int* ev_k; R* ev_c;            // [B][cap]
    int* slot_t; int* slot_k; double* slot_a; // [B][cap]
    unsigned long long* hkey; int* hval;      // [B][hmask+1] open-addressing table over the slots (built by the loop once a signal has many)
    int* sel_t; int* sel_k; R* sel_c;         // [B][2*maxsel]  (two halves: raw / ordered)
    int* stats;         // [B][ST_COUNT]
    R* energy;          // [B][2]: signal, residual
    unsigned long long* edge;   // [B][kEdgeWords]: edge rows re-correlated at least once + the stale-sample record (score-only policies)
    int* head;          // [B][T] round-parallel loop: most recent coefficient slot at position t (-1: none), chained through hval
    double* lgram;      // [B][lgram_doubles(lg_cap)] LoCOMP: what a group keeps beyond its LDS copy (nullptr: other methods)
};

__device__ __forceinline__ float rabs(float v) { return fabsf(v); }
__device__ __forceinline__ double rabs(double v) { return fabs(v); }
__device__ __forceinline__ float rfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double rfma(double a, double b, double c) { return fma(a, b, c); }

// |c * w_k| exactly as modeling.py:906 then np.abs (one rounded product), or |c| without weights
template <typename R> __device__ __forceinline__ R score_of(R c, int k, const R* w)
{
    if (w) { R s = c * w[k]; return rabs(s); }
    return rabs(c);
}

// (score, index) candidate; larger score wins, then the smaller index (C-order arg-max, :967)
template <typename R> struct Cand { R s; int i; };
template <typename R> __device__ __forceinline__ bool better(const Cand<R>& a, const Cand<R>& b)
{
    return a.s > b.s || (a.s == b.s && a.i < b.i);
}
template <typename R> __device__ __forceinline__ Cand<R> wave_argmax(Cand<R> c)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Cand<R> o;
        o.s = __shfl_xor(c.s, m);
        o.i = __shfl_xor(c.i, m);
        if (better(o, c)) c = o;
    }
    return c;
}

// float specialisation on DPP (data-parallel primitives: VALU-speed lane exchanges instead of the
// LDS-crossbar ds_bpermute behind __shfl_xor): quad swaps, half-row and row mirrors bring every
// 16-lane row to its own arg-max -- `better` is symmetric, so mirrored partners are as good as xor
// partners -- then the four row results are read out with v_readlane and merged.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ void argmax_step(Cand<float>& c)
{
    Cand<float> o;
    o.s = dpp_f<CTRL>(c.s);
    o.i = dpp_i<CTRL>(c.i);
    if (better(o, c)) c = o;
}
// Scores are >= 0 (or -1 for "nothing"), so the arg-max splits into two plain reductions, each ONE fused DPP
// instruction per step (v_max_f32_dpp / v_min_i32_dpp: the vector ALU is the resource the f32 MFMA shares, so
// instruction count is what matters here): the wave maximum of the score, then the smallest index among the lanes that
// hold it.  Steps: quad swaps, half-row and row mirrors (every row of 16 holds its result), row_bcast 15 / 31 (rows
// 1..3 fold in their predecessors: lane 63 holds the wave's result), v_readlane.  Same winner as `better` picks.
// (hand-written: the compiler's form of a DPP reduction step is v_mov + v_mov_dpp + v_max -- three to four vector
// instructions -- where one fused v_max_i32_dpp does; s_nop 1 covers the VALU-write -> DPP-read hazard.  Scores are
// compared through their bit patterns: non-negative floats order like integers, and the "nothing" sentinel -1.0f is
// a negative integer.)
__device__ __forceinline__ int wave_max_i32(int v)          // result in every lane (wave-uniform)
{
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v)
{
    asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
template <> __device__ __forceinline__ Cand<float> wave_argmax<float>(Cand<float> c)
{
    Cand<float> r;
    const int sb = __float_as_int(c.s);
    const int mb = wave_max_i32(sb);
    r.s = __int_as_float(mb);
    r.i = wave_min_i32(sb == mb ? c.i : INT_MAX);
    return r;
}

// Arg-max for candidates that are ORDERED BY LANE (every index a lane may hold is below every index of the next lane):
// the first lane holding the maximal score holds the lowest index, so the second reduction of wave_argmax (minimum index
// among the maxima: compare, select, six DPP steps, readlane -- all dependent) becomes a ballot, s_ff1 and one readlane.
// Dependent vector instructions are what a wave waits for beside other waves' matrix instructions
// (tools/serial_stretch_probe.hip), so the selection scans hand out contiguous index ranges per lane to qualify.
template <typename R> __device__ __forceinline__ Cand<R> wave_argmax_first(Cand<R> c) { return wave_argmax(c); }
template <> __device__ __forceinline__ Cand<float> wave_argmax_first<float>(Cand<float> c)
{
    Cand<float> r;
    const int sb = __float_as_int(c.s);
    const int mb = wave_max_i32(sb);
    const unsigned long long holders = __ballot(sb == mb);
    r.s = __int_as_float(mb);
    r.i = __builtin_amdgcn_readlane(c.i, __ffsll((long long)holders) - 1);
    return r;
}

template <typename R> __device__ __forceinline__ R wave_max(R v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { R o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}

// Pinned summation tree (DESIGN.md "Numerics", oracle hsco_energy_*): the caller has formed the
// 256 strided partials p[tid]; halving tree inside each wave, then (P0+P1)+(P2+P3).
// Returns the total in thread 0 (other threads: unspecified).  `scratch` holds >= 2*kWaves R's.
// Halving tree of the pinned sums inside one wave: lane i += lane i+m for m = 32, 16, 8, 4, 2, 1; the total ends in
// lane 0 (the other lanes hold partial garbage nobody reads).  float: the partner comes through v_permlane32_swap /
// v_permlane16_swap and row_shl DPP adds instead of six ds_bpermute round trips; same operands, same order.
__device__ __forceinline__ void wave_tree_down2(float& a, float& b)
{
    {
        const auto ra = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(a), false, false);
        const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_int(b), __float_as_int(b), false, false);
        a = a + __int_as_float((int)ra[1]);
        b = b + __int_as_float((int)rb[1]);
    }
    {
        const auto ra = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(a), false, false);
        const auto rb = __builtin_amdgcn_permlane16_swap(__float_as_int(b), __float_as_int(b), false, false);
        a = a + __int_as_float((int)ra[1]);
        b = b + __int_as_float((int)rb[1]);
    }
#define HSCMP_TREE_STEP(CTRL)                                                                                                   \
    a = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, true));                            \
    b = b + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), CTRL, 0xF, 0xF, true));
    HSCMP_TREE_STEP(0x108)      // row_shl:8
    HSCMP_TREE_STEP(0x104)
    HSCMP_TREE_STEP(0x102)
    HSCMP_TREE_STEP(0x101)
#undef HSCMP_TREE_STEP
}
__device__ __forceinline__ void wave_tree_down2(double& a, double& b)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double oa = __shfl_down(a, m), ob = __shfl_down(b, m);
        a = a + oa;
        b = b + ob;
    }
}

template <typename R, typename SY> __device__ __forceinline__ void pinned_tree2(R& a, R& b, R* scratch, SY& sy)
{
    wave_tree_down2(a, b);
    const int lane = ltid() & 63, wv = ltid() >> 6;
    sy.lds();                                           // (scratch lives in LDS: no need to drain global stores here)
    if (lane == 0) { scratch[wv] = a; scratch[kWaves + wv] = b; }
    sy.lds();
    if (ltid() == 0) {
        R a01 = scratch[0] + scratch[1];
        R a23 = scratch[2] + scratch[3];
        a = a01 + a23;
        R b01 = scratch[kWaves + 0] + scratch[kWaves + 1];
        R b23 = scratch[kWaves + 2] + scratch[kWaves + 3];
        b = b01 + b23;
    }
}
template <typename R> __device__ __forceinline__ void pinned_tree2(R& a, R& b, R* scratch)
{
    HwSync hw;
    pinned_tree2(a, b, scratch, hw);
}

// utils.py:76-161: clipped support [s,e) of a width-W element centred at t, element offset es
__device__ __forceinline__ int centered_span(int T, int W, int t, int& s, int& e, int& es)
{
    const int lo = t - (W - 1) / 2;
    const int hi = t + W / 2 + 1;
    s = lo < 0 ? 0 : lo;
    e = hi > T ? T : hi;
    es = s - lo;
    return e - s;
}

// np.pad(mode='reflect') of the slice [sidx, sidx+n) evaluated at global index g (modeling.py:1046)
__device__ __forceinline__ int reflect_index(int g, int sidx, int n)
{
    if (n == 1) return sidx;
    const int period = 2 * (n - 1);
    int m = (g - sidx) % period;
    if (m < 0) m += period;
    if (m >= n) m = period - m;
    return sidx + m;
}

}  // namespace hscmp
