// hscmp_api.hip -- host side of libhscmp.so (C ABI declared in include/hscmp.h).
//
// Owns the GPU-resident dictionary and the per-batch workspace (table-free state, event lists),
// translates the reference's keyword arguments (hsc/modeling.py:1053) into kernel parameters and
// queues prepare -> initial correlation -> greedy loop on one HIP stream.  No CPU compute path:
// every entry point either runs the HIP kernels or fails with an error code.
#include "../../include/hscmp.h"

#include "hscmp_kernels.h"
#include "hscmp_mfma.h"
#include "hscmp_sparse.h"
#include "hscmp_rp.h"
#include "hscmp_rp_sparse.h"
#include "hscmp_locomp.h"
#include "hscmp_epilogue.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

using namespace hscmp;

struct hscmp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    std::string variant = "none";
    // dictionary
    int K = 0, W = 0, F = 0, dtype = -1;
    void* d_D = nullptr;
    void* d_w = nullptr;      // nullptr when no weights
    void* d_Dfrag = nullptr;  // MFMA fragment-ordered copy (f32, F == 1)
    void* d_Dt = nullptr;     // [W][F][K] transposed copy for the sparsity-aware kernels (F > 1)
    void* d_Dc = nullptr;     // [K][F][W] chain-ordered copy for the dense chains (F > 1)
    void* d_scratch = nullptr;
    int* d_nzptr = nullptr;   // CSR of the dictionary's non-zeros per atom, chain order (sparse level dictionaries)
    int* d_nzwf = nullptr;
    void* d_nzval = nullptr;
    int dict_nnz = 0;
    int* d_fptr = nullptr;    // the same non-zeros grouped by feature
    int* d_fkw = nullptr;
    void* d_fval = nullptr;
    int* d_rl_cnt = nullptr;  // per-row feature lists of the residual's possibly non-zero cells (sparse dictionaries)
    int* d_rl_f = nullptr;
    bool rl_filled = false;   // the lists of the current input were written by the level chaining
    // rows of d_resid (x F float64) that a chained encode left behind with every possibly non-zero cell named by
    // d_rl_cnt / d_rl_f: the next chained encode clears those cells instead of the whole buffer (0: clear everything)
    int64_t listed_rows = 0;
    int listed_F = 0;
    bool loop_kept_lists = false;   // the last encode ran the sparse loop with row lists (it enters every cell it writes)
    unsigned char* d_rowflag = nullptr;  // [B][T] non-zero input rows handed over by the level chaining
    bool rowflag_valid = false;
    size_t Dfrag_bytes = 0;
    // batch workspace
    int B = 0, T = 0, cap = 0, maxsel = 0;
    bool have_batch = false;
    size_t caps[16] = {0};
    size_t cap_scratch = 0, cap_rowflag = 0, cap_rl_cnt = 0, cap_rl_f = 0;
    void* d_x = nullptr;      // staging for host inputs
    void* d_resid = nullptr; void* d_best_c = nullptr; int* d_best_k = nullptr;
    int* d_ev_t = nullptr; int* d_ev_k = nullptr; void* d_ev_c = nullptr;
    int* d_slot_t = nullptr; int* d_slot_k = nullptr; double* d_slot_a = nullptr;
    unsigned long long* d_hkey = nullptr; int* d_hval = nullptr;   // slot hash table [B][hmask+1]
    int* d_head = nullptr;     // [B][T] slot chains by position (round-parallel loop)
    double* d_lgram = nullptr; // [B][kLgramDoubles] LoCOMP: Gram matrices beyond the LDS copy
    size_t cap_hkey = 0, cap_hval = 0, cap_head = 0, cap_lgram = 0;
    int* d_sel_t = nullptr; int* d_sel_k = nullptr; void* d_sel_c = nullptr;
    int* d_stats = nullptr; void* d_energy = nullptr; unsigned long long* d_edge = nullptr;
    DevParams P{};
    hscmp_params last{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timed = false;
    bool timed_loop_only = false;   // the last timed launch was a hscmp_continue (no prepare / initial correlation)
    bool mfma_state = false;        // the batch's table-free state is the score-only form of the MFMA kernels
    bool rp_last = false;           // the last loop launch was the round-parallel form (hscmp_rp.h)
    int method = 0;                 // hscmp_set_method: 0 = greedy pursuit (modeling.py:1053), 1 = LoCOMP (:1267)
    bool locomp_state = false;      // the batch was encoded by the LoCOMP loop (hscmp_continue resumes it)
    bool locomp_sparse = false;     // ... on the sparse policy (LocompSparse)
    bool locomp_mfma = false;       // ... with the re-correlations on the matrix cores (LocompMfma)
    const void* last_x_dev = nullptr;   // device address of the signals of the last encode (hscmp_hierarchy_epilogue reads them)
    // workspace arena of the entry points outside the batch encode (grow-only, lives as long as the context): slots
    // 0-7 the hierarchical epilogue, 8-15 the row-level entry points and the device-resident table
    void* d_epi[16] = {nullptr};
    size_t cap_epi[16] = {0};
    // device-resident inner-product table of LoCOMP (hscmp_table_*): [T][K] in slot kArenaTable, its residual in kArenaTabRes
    int tab_T = 0;
};
enum { kArenaRowA = 8, kArenaRowB = 9, kArenaRowC = 10, kArenaRowD = 11, kArenaTable = 12, kArenaTabRes = 13, kArenaTabW = 14, kArenaEpiEnergy = 15 };

static thread_local std::string g_err;

static int fail(hscmp_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail(ctx, HSCMP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

static size_t esize(int dtype) { return dtype == HSCMP_F64 ? 8 : 4; }

// Arena slot i holds at least `bytes` afterwards.  Grow-only: a call that fits reuses the buffer (no hipMalloc in the
// steady state of any entry point); a call that does not fit waits for the stream (kernels may still read the old
// buffer), frees and allocates 1/8 more than asked.  `keep`: the old contents are copied over.
static int epi_buffer(hscmp_ctx* ctx, int i, size_t bytes, bool keep = false)
{
    if (ctx->d_epi[i] && ctx->cap_epi[i] >= bytes) return HSCMP_OK;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, HSCMP_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    const size_t want = bytes + bytes / 8 + 256;
    void* fresh = nullptr;
    e = hipMalloc(&fresh, want);
    if (e != hipSuccess) return fail(ctx, HSCMP_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    if (ctx->d_epi[i]) {
        if (keep && (e = hipMemcpy(fresh, ctx->d_epi[i], ctx->cap_epi[i], hipMemcpyDeviceToDevice)) != hipSuccess) {
            (void)hipFree(fresh);
            return fail(ctx, HSCMP_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
        }
        (void)hipFree(ctx->d_epi[i]);
    }
    ctx->d_epi[i] = fresh; ctx->cap_epi[i] = want;
    return HSCMP_OK;
}

extern "C" int hscmp_version(void) { return HSCMP_VERSION; }

extern "C" const char* hscmp_last_error(hscmp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

extern "C" int hscmp_create(hscmp_ctx** out, int device_id)
{
    if (!out) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, HSCMP_ERR_NO_DEVICE, "hscmp_create: no HIP device visible (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_create: device %d out of range (%d devices)", device_id, n);
    hscmp_ctx* ctx = new hscmp_ctx();
    ctx->device = device_id;
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&ctx->ev[i]);
    if (e != hipSuccess) {
        int rc = fail(nullptr, HSCMP_ERR_HIP, "hscmp_create: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return HSCMP_OK;
}

static void free_all(hscmp_ctx* c)
{
    for (void* p : c->d_epi) if (p) (void)hipFree(p);
    void* ptrs[] = {c->d_D, c->d_w, c->d_Dfrag, c->d_Dt, c->d_Dc, c->d_nzptr, c->d_nzwf, c->d_nzval, c->d_fptr, c->d_fkw, c->d_fval, c->d_rl_cnt, c->d_rl_f, c->d_scratch, c->d_rowflag, c->d_x, c->d_resid, c->d_best_c, c->d_best_k, c->d_ev_t, c->d_ev_k, c->d_ev_c,
                    c->d_slot_t, c->d_slot_k, c->d_slot_a, c->d_hkey, c->d_hval, c->d_head, c->d_lgram, c->d_sel_t, c->d_sel_k, c->d_sel_c, c->d_stats, c->d_energy, c->d_edge};
    for (void* p : ptrs) if (p) (void)hipFree(p);
}

extern "C" void hscmp_destroy(hscmp_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_all(ctx);
    for (int i = 0; i < 4; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int hscmp_set_method(hscmp_ctx* ctx, int method)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_set_method: ctx is NULL");
    if (method != HSCMP_METHOD_CMP && method != HSCMP_METHOD_LOCOMP) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_set_method: unknown method %d", method);
    ctx->method = method;
    return HSCMP_OK;
}

extern "C" int hscmp_set_stream(hscmp_ctx* ctx, void* hip_stream)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_set_stream: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return HSCMP_OK;
}

extern "C" int hscmp_synchronize(hscmp_ctx* ctx)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_synchronize: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_set_dictionary(hscmp_ctx* ctx, const void* D, int K, int W, int F, hscmp_dtype dtype, const void* weights)
{
    if (ctx) { ctx->tab_T = 0; ctx->listed_rows = 0; }     // a resident table belongs to the dictionary it was built with
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_set_dictionary: ctx is NULL");
    if (!D || K <= 0 || W <= 0 || F <= 0) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_set_dictionary: bad shape K=%d W=%d F=%d", K, W, F);
    if (dtype != HSCMP_F32 && dtype != HSCMP_F64) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_set_dictionary: bad dtype %d", (int)dtype);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t es = esize(dtype);
    const size_t nD = (size_t)K * W * F * es;
    if (ctx->d_D) { (void)hipFree(ctx->d_D); ctx->d_D = nullptr; }
    if (ctx->d_w) { (void)hipFree(ctx->d_w); ctx->d_w = nullptr; }
    if (ctx->d_Dfrag) { (void)hipFree(ctx->d_Dfrag); ctx->d_Dfrag = nullptr; ctx->Dfrag_bytes = 0; }
    if (ctx->d_Dt) { (void)hipFree(ctx->d_Dt); ctx->d_Dt = nullptr; }
    if (ctx->d_Dc) { (void)hipFree(ctx->d_Dc); ctx->d_Dc = nullptr; }
    if (ctx->d_nzptr) { (void)hipFree(ctx->d_nzptr); ctx->d_nzptr = nullptr; }
    if (ctx->d_nzwf) { (void)hipFree(ctx->d_nzwf); ctx->d_nzwf = nullptr; }
    if (ctx->d_nzval) { (void)hipFree(ctx->d_nzval); ctx->d_nzval = nullptr; }
    if (ctx->d_fptr) { (void)hipFree(ctx->d_fptr); ctx->d_fptr = nullptr; }
    if (ctx->d_fkw) { (void)hipFree(ctx->d_fkw); ctx->d_fkw = nullptr; }
    if (ctx->d_fval) { (void)hipFree(ctx->d_fval); ctx->d_fval = nullptr; }
    HIP_TRY(ctx, hipMalloc(&ctx->d_D, nD));
    HIP_TRY(ctx, hipMemcpy(ctx->d_D, D, nD, hipMemcpyHostToDevice));
    // weights that are all exactly 1 select like no weights at all (|c * 1| == |c| bit for bit; the level-0 weights of the
    // hierarchical encoder, modeling.py:1448-1450 with no singletons): the unweighted kernels are the cheaper instances
    if (weights) {
        bool all_one = true;
        for (int k = 0; k < K && all_one; ++k)
            all_one = dtype == HSCMP_F32 ? ((const float*)weights)[k] == 1.0f : ((const double*)weights)[k] == 1.0;
        if (all_one) weights = nullptr;
    }
    if (weights) {
        HIP_TRY(ctx, hipMalloc(&ctx->d_w, (size_t)K * es));
        HIP_TRY(ctx, hipMemcpy(ctx->d_w, weights, (size_t)K * es, hipMemcpyHostToDevice));
    }
    ctx->K = K; ctx->W = W; ctx->F = F; ctx->dtype = dtype;
    ctx->have_batch = false;
    if (F > 1) {
        // Dt[w][f][k] = D[k][w][f]: atom index contiguous, for the gathered-window kernels
        std::vector<char> dt(nD);
        for (int k = 0; k < K; ++k)
            for (int w = 0; w < W; ++w)
                for (int f = 0; f < F; ++f)
                    memcpy(&dt[(((size_t)w * F + f) * K + k) * es], (const char*)D + (((size_t)k * W + w) * F + f) * es, es);
        HIP_TRY(ctx, hipMalloc(&ctx->d_Dt, nD));
        HIP_TRY(ctx, hipMemcpy(ctx->d_Dt, dt.data(), nD, hipMemcpyHostToDevice));
        // Dc[k][f][w] = D[k][w][f]: consecutive addresses along the pinned chain (f outer, w inner)
        for (int k = 0; k < K; ++k)
            for (int w = 0; w < W; ++w)
                for (int f = 0; f < F; ++f)
                    memcpy(&dt[(((size_t)k * F + f) * W + w) * es], (const char*)D + (((size_t)k * W + w) * F + f) * es, es);
        HIP_TRY(ctx, hipMalloc(&ctx->d_Dc, nD));
        HIP_TRY(ctx, hipMemcpy(ctx->d_Dc, dt.data(), nD, hipMemcpyHostToDevice));
        // per-atom list of non-zeros in chain order (f outer, w inner), kept when the dictionary is sparse
        // (level dictionaries built from decompositions + singletons, hsc/dataset.py:137-194, 826-860)
        if (W <= 32767 && F <= 65535 && !getenv("HSCMP_NO_DICT_LISTS")) {
            std::vector<int> ptr(K + 1, 0), wf;
            std::vector<char> val;
            const size_t limit = (size_t)kDictListMaxPerAtom * K;
            bool sparse = true;
            for (int k = 0; k < K && sparse; ++k) {
                for (int f = 0; f < F && sparse; ++f)
                    for (int w = 0; w < W; ++w) {
                        const char* src = (const char*)D + (((size_t)k * W + w) * F + f) * es;
                        const bool nz = dtype == HSCMP_F32 ? (*(const float*)src != 0.0f) : (*(const double*)src != 0.0);
                        if (!nz) continue;
                        if (wf.size() >= limit) { sparse = false; break; }
                        wf.push_back((w << 16) | f);
                        val.insert(val.end(), src, src + es);
                    }
                ptr[k + 1] = (int)wf.size();
            }
            if (sparse) {
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_nzptr, (K + 1) * sizeof(int)));
                HIP_TRY(ctx, hipMemcpy(ctx->d_nzptr, ptr.data(), (K + 1) * sizeof(int), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_nzwf, std::max<size_t>(1, wf.size()) * sizeof(int)));
                HIP_TRY(ctx, hipMemcpy(ctx->d_nzwf, wf.data(), wf.size() * sizeof(int), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMalloc(&ctx->d_nzval, std::max<size_t>(es, val.size())));
                HIP_TRY(ctx, hipMemcpy(ctx->d_nzval, val.data(), val.size(), hipMemcpyHostToDevice));
                // grouped by feature: counting sort of the per-atom lists
                if (K <= 65535) {
                    const size_t nnz = wf.size();
                    ctx->dict_nnz = (int)nnz;
                    std::vector<int> fp(F + 1, 0), kw(nnz);
                    std::vector<char> fv(std::max<size_t>(es, nnz * es));
                    for (size_t e = 0; e < nnz; ++e) fp[(wf[e] & 0xffff) + 1] += 1;
                    for (int f = 0; f < F; ++f) fp[f + 1] += fp[f];
                    std::vector<int> cur(fp.begin(), fp.end() - 1);
                    for (int k = 0; k < K; ++k)
                        for (int e = ptr[k]; e < ptr[k + 1]; ++e) {
                            const int o = cur[wf[e] & 0xffff]++;
                            kw[o] = (int)(((unsigned)k << 16) | (unsigned)(wf[e] >> 16));
                            memcpy(&fv[(size_t)o * es], &val[(size_t)e * es], es);
                        }
                    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_fptr, (F + 1) * sizeof(int)));
                    HIP_TRY(ctx, hipMemcpy(ctx->d_fptr, fp.data(), (F + 1) * sizeof(int), hipMemcpyHostToDevice));
                    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_fkw, std::max<size_t>(1, nnz) * sizeof(int)));
                    HIP_TRY(ctx, hipMemcpy(ctx->d_fkw, kw.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
                    HIP_TRY(ctx, hipMalloc(&ctx->d_fval, fv.size()));
                    HIP_TRY(ctx, hipMemcpy(ctx->d_fval, fv.data(), fv.size(), hipMemcpyHostToDevice));
                }
            }
        }
    }
    // MFMA operand image of the dictionary (f32 only): built once, reused by every encode
    if (dtype == HSCMP_F32 && mfma_supported<float>(K, W, F)) {
        std::vector<float> frag;
        mfma_build_dict_image((const float*)D, K, W, F, frag);
        ctx->Dfrag_bytes = frag.size() * sizeof(float);
        HIP_TRY(ctx, hipMalloc(&ctx->d_Dfrag, ctx->Dfrag_bytes));
        HIP_TRY(ctx, hipMemcpy(ctx->d_Dfrag, frag.data(), ctx->Dfrag_bytes, hipMemcpyHostToDevice));
    } else if (dtype == HSCMP_F64 && mfma_supported<double>(K, W, F)) {
        std::vector<double> frag;
        mfma_build_dict_image_f64((const double*)D, K, W, frag);
        ctx->Dfrag_bytes = frag.size() * sizeof(double);
        HIP_TRY(ctx, hipMalloc(&ctx->d_Dfrag, ctx->Dfrag_bytes));
        HIP_TRY(ctx, hipMemcpy(ctx->d_Dfrag, frag.data(), ctx->Dfrag_bytes, hipMemcpyHostToDevice));
    }
    return HSCMP_OK;
}

// geometry given explicitly (the row-level selection runs on a caller's table of any K, W: the context's dictionary is
// not involved and is not touched)
static int make_params_g(hscmp_ctx* ctx, int K, int W, int F, int B, int T, const hscmp_params* p, DevParams* out)
{
    DevParams P{};
    P.B = B; P.T = T; P.K = K; P.W = W; P.F = F;
    P.off = (W - 1) / 2;
    set_segments(P, kMaxSeg);
    if (p->nb_blocks == 1) { P.blocked = 0; P.bs = 0; P.nbk = 0; P.maxsel = 1; }
    else {
        // modeling.py:908-918
        int bs;
        if (p->nb_blocks < 0) bs = 4 * W;
        else if (p->nb_blocks > 1) bs = (int)std::floor((double)T / (double)p->nb_blocks);
        else return fail(ctx, HSCMP_ERR_INVALID, "nb_blocks must be 1, > 1 or -1 ('auto'), got %d", p->nb_blocks);
        if (bs % 2 == 1) bs += 1;
        if (bs <= 0) return fail(ctx, HSCMP_ERR_INVALID, "nbBlocks=%d gives an empty block for T=%d", p->nb_blocks, T);
        P.blocked = 1; P.bs = bs; P.nbk = (int)std::ceil((double)T / (double)bs); P.maxsel = P.nbk + 1;
    }
    P.l0 = p->nb_nonzero_coefs < 0 ? -1 : p->nb_nonzero_coefs;
    P.has_snr = !std::isnan(p->tolerance_snr);
    P.has_scale = !std::isnan(p->tolerance_residual_scale);
    P.has_thres = !std::isnan(p->null_coeff_thres);
    P.snr_ratio = P.has_snr ? std::pow(10.0, p->tolerance_snr / 10.0) : 0.0;
    P.tol_scale = P.has_scale ? p->tolerance_residual_scale : 0.0;
    P.thres = P.has_thres ? p->null_coeff_thres : 0.0;
    P.eps = p->eps;
    if (p->max_events <= 0) return fail(ctx, HSCMP_ERR_INVALID, "max_events must be > 0");
    P.cap = p->max_events;
    P.hmask = slot_hash_mask(P.cap);
    P.hash_min = kSlotHashMin;
    if (const char* v = getenv("HSCMP_SLOT_HASH_MIN")) P.hash_min = std::max(0, atoi(v));
    if (ctx && ctx->method == HSCMP_METHOD_LOCOMP) P.hash_min = INT_MAX;        // (its atom body scans the slot list for the neighbourhood anyway)
    P.max_rounds = p->max_rounds;
    P.lg_cap = kLocompGroupCap;
    if (const char* v = getenv("HSCMP_LOCOMP_GROUP_CAP")) P.lg_cap = std::min(4096, std::max(2, atoi(v)));
    P.lc_ahead = 7;          // bit 0: selections of a round side by side; bit 1: a group's rows re-correlated one wave per quarter (sparse policy);
                             // bit 2: the rows of a batch of selections re-correlated behind its last one, one wave per selection
    if (const char* v = getenv("HSCMP_LOCOMP_AHEAD")) P.lc_ahead = atoi(v) & 7;
    *out = P;
    return HSCMP_OK;
}

static int make_params(hscmp_ctx* ctx, int B, int T, const hscmp_params* p, DevParams* out)
{
    return make_params_g(ctx, ctx->K, ctx->W, ctx->F, B, T, p, out);
}

// every buffer tracks its own capacity in bytes (element size changes with the dictionary dtype)
struct BufCap { void** p; size_t* cap; size_t bytes; };

// Per-row feature lists: multi-feature inputs with a sparse dictionary (the per-atom lists tell which cells an
// atom touches).
constexpr int kRowListCap = 8;
static bool use_row_lists(const hscmp_ctx* ctx)
{
    return ctx->F > 1 && ctx->d_nzptr != nullptr && !getenv("HSCMP_NO_ROW_LISTS") && !getenv("HSCMP_FORCE_DENSE");
}

// Workgroups per signal of the sparse initial correlation: enough to fill the chip at small batches.
static int sparse_init_split(int B, int T, int W)
{
    const int nblocks = (T + 2 * W - 2) / (2 * W - 1);
    return std::max(1, std::min(nblocks, (2048 + B - 1) / B));
}

static int ensure_workspace_g(hscmp_ctx* ctx, const DevParams& P, bool need_x, size_t es, bool multi_feature, bool row_lists);
static int ensure_workspace(hscmp_ctx* ctx, const DevParams& P, bool need_x)
{
    return ensure_workspace_g(ctx, P, need_x, esize(ctx->dtype), ctx->F > 1, use_row_lists(ctx));
}
// (element size and feature layout given explicitly: see make_params_g)
static int ensure_workspace_g(hscmp_ctx* ctx, const DevParams& P, bool need_x, size_t es, bool multi_feature, bool row_lists)
{
    const size_t B = P.B, TF = (size_t)P.T * P.F, T = P.T, cap = P.cap, ms = P.maxsel;
    BufCap bufs[] = {
        {(void**)&ctx->d_x, &ctx->caps[0], need_x ? B * TF * es : 0},
        {(void**)&ctx->d_resid, &ctx->caps[1], B * TF * es},
        {(void**)&ctx->d_best_c, &ctx->caps[2], B * T * es},
        {(void**)&ctx->d_best_k, &ctx->caps[3], B * T * sizeof(int)},
        {(void**)&ctx->d_ev_t, &ctx->caps[4], B * cap * 4},
        {(void**)&ctx->d_ev_k, &ctx->caps[5], B * cap * 4},
        {(void**)&ctx->d_ev_c, &ctx->caps[6], B * cap * es},
        {(void**)&ctx->d_slot_t, &ctx->caps[7], B * cap * 4},
        {(void**)&ctx->d_slot_k, &ctx->caps[8], B * cap * 4},
        {(void**)&ctx->d_slot_a, &ctx->caps[9], B * cap * 8},
        {(void**)&ctx->d_sel_t, &ctx->caps[10], B * 2 * ms * 4},
        {(void**)&ctx->d_sel_k, &ctx->caps[11], B * 2 * ms * 4},
        {(void**)&ctx->d_sel_c, &ctx->caps[12], B * 2 * ms * es},
        {(void**)&ctx->d_stats, &ctx->caps[13], B * ST_COUNT * sizeof(int)},
        {(void**)&ctx->d_energy, &ctx->caps[14], B * 2 * es},
        {(void**)&ctx->d_edge, &ctx->caps[15], B * kEdgeWords * sizeof(unsigned long long)},
        {(void**)&ctx->d_scratch, &ctx->cap_scratch, multi_feature ? B * (size_t)sparse_init_split(P.B, P.T, P.W) * (2 * P.W - 1) * P.K * es : 0},
        {(void**)&ctx->d_rowflag, &ctx->cap_rowflag, multi_feature ? B * T : 0},
        {(void**)&ctx->d_rl_cnt, &ctx->cap_rl_cnt, row_lists ? B * T * sizeof(int) : 0},
        {(void**)&ctx->d_rl_f, &ctx->cap_rl_f, row_lists ? B * T * kRowListCap * sizeof(int) : 0},
        {(void**)&ctx->d_hkey, &ctx->cap_hkey, B * ((size_t)P.hmask + 1) * sizeof(unsigned long long)},
        {(void**)&ctx->d_hval, &ctx->cap_hval, B * ((size_t)P.hmask + 1) * sizeof(int)},
        {(void**)&ctx->d_head, &ctx->cap_head, (P.blocked || ctx->method == HSCMP_METHOD_LOCOMP) ? B * T * sizeof(int) : 0},
        {(void**)&ctx->d_lgram, &ctx->cap_lgram, ctx->method == HSCMP_METHOD_LOCOMP ? B * lgram_doubles(P.lg_cap) * sizeof(double) : 0},
    };
    bool stream_idle = false;
    for (const BufCap& b : bufs) {
        if (b.bytes == 0 || (*b.p && *b.cap >= b.bytes)) continue;
        if (!stream_idle) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); stream_idle = true; }
        if (*b.p) { (void)hipFree(*b.p); *b.p = nullptr; *b.cap = 0; }
        ctx->listed_rows = 0;                   // (a fresh buffer knows nothing of the previous batch)
        hipError_t e = hipMalloc(b.p, b.bytes);
        if (e != hipSuccess) return fail(ctx, HSCMP_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", b.bytes, hipGetErrorString(e));
        *b.cap = b.bytes;
    }
    return HSCMP_OK;
}

template <typename R> static State<R> make_state(hscmp_ctx* c)
{
    State<R> S;
    S.D = (const R*)c->d_D; S.weights = (const R*)c->d_w;
    S.Dc = c->d_Dc ? (const R*)c->d_Dc : (const R*)c->d_D;
    S.residual = (R*)c->d_resid; S.best_c = (R*)c->d_best_c; S.best_k = c->d_best_k;
    S.ev_t = c->d_ev_t; S.ev_k = c->d_ev_k; S.ev_c = (R*)c->d_ev_c;
    S.slot_t = c->d_slot_t; S.slot_k = c->d_slot_k; S.slot_a = c->d_slot_a;
    S.hkey = c->d_hkey; S.hval = c->d_hval; S.head = c->d_head; S.lgram = c->d_lgram;
    S.sel_t = c->d_sel_t; S.sel_k = c->d_sel_k; S.sel_c = (R*)c->d_sel_c;
    S.stats = c->d_stats; S.energy = (R*)c->d_energy; S.edge = c->d_edge;
    return S;
}

static bool use_mfma(const hscmp_ctx* ctx, int T)
{
    if (getenv("HSCMP_FORCE_GENERIC")) return false;
    // the score-only MFMA path assumes single-bounce reflection at the edges (T >= 3W-2)
    return ctx->d_Dfrag != nullptr && T >= 3 * ctx->W - 2;
}

// Round-parallel loop (hscmp_rp.h: a 1024-thread workgroup per signal, the atoms of a blocked round side by side).
// HSCMP_RP=0/1 forces the choice (tests run both; the results are bit-identical).
static bool use_rp(const DevParams& P, bool level_loop)
{
    if (!P.blocked) return false;
    if (const char* e = getenv("HSCMP_RP")) return atoi(e) != 0;
    // measured at the config-4 shape (profiles/r03_*): the level loops gain at every batch size (1024 signals: 28.9 ->
    // 14.5 ms); on the matrix cores the four-signal loop catches up once every CU holds four signals (1024: 117.2 vs 117.6 ms;
    // 512: 66.0 vs 60.7 ms)
    return level_loop || P.B <= 3 * mfma_device_cus();
}

// the MFMA loop of a float32 batch: round-parallel when the batch is small and the round is blocked, else iterate_kernel
template <typename R> static int launch_mfma_loop(hscmp_ctx* ctx, const DevParams& P, const State<R>& S)
{
    ctx->rp_last = false;
    if constexpr (sizeof(R) == 4) {
        if (use_rp(P, false) && rp_mfma_launch(ctx->stream, P, S, (const float*)ctx->d_Dfrag, true) == 0) {
            if (rp_mfma_launch(ctx->stream, P, S, (const float*)ctx->d_Dfrag) != 0) return -1;
            ctx->rp_last = true;
            return 0;
        }
    }
    return mfma_launch_iterate<R>(ctx->stream, P, S, (const R*)ctx->d_Dfrag);
}

// Loop policy for multi-feature inputs (hierarchical levels >= 1): SparseRecorr gathers the non-zeros
// of the window into LDS and reads the transposed dictionary coalesced.
static bool use_sparse_loop(const hscmp_ctx* ctx)
{
    if (getenv("HSCMP_FORCE_DENSE")) return false;
    // multi-feature inputs only (measured: for dense single-feature windows the dense chain is 3x faster), and
    // only with a sparse dictionary: subtracting dense atoms fills the residual, the windows then overflow the
    // gathered lists and the dense LDS-staged chain of GenericRecorr is several times faster (a k-means
    // dictionary with ~150 of 528 non-zeros per atom: 1.3 ms vs 0.37 ms per atom)
    if (ctx->d_nzptr == nullptr && !getenv("HSCMP_FORCE_GATHERED")) return false;
    return ctx->F > 1 && ctx->d_Dt != nullptr && ctx->W <= 16384 && ctx->F <= 32767;   // (f << 16) | row key
}
// Sparse INITIAL correlation: only for multi-feature inputs (hierarchical levels >= 1, almost all
// zero); a dense single-feature signal is cheaper through the dense generic kernel.
static bool use_sparse_init(const hscmp_ctx* ctx, int T)
{
    if (getenv("HSCMP_FORCE_DENSE")) return false;
    return ctx->F > 1 && ctx->d_Dt != nullptr && ctx->W <= 16384 && ctx->F <= 32767 && T <= 262144;
}

template <typename R> static SparseArgs<R> sparse_args(hscmp_ctx* ctx, int T, bool packed = false)
{
    SparseArgs<R> A;
    A.Dt = (const R*)ctx->d_Dt; A.scratch = (R*)ctx->d_scratch;
    A.rowflag = (T <= kRowBitsMaxT && !getenv("HSCMP_NO_ROWBITS")) ? ctx->d_rowflag : nullptr;
    A.rowflag_filled = ctx->rowflag_valid ? 1 : 0;
    A.nzptr = ctx->d_nzptr; A.nzwf = ctx->d_nzwf; A.nzval = (const R*)ctx->d_nzval;
    A.fptr = getenv("HSCMP_NO_PAIRING") ? nullptr : ctx->d_fptr; A.fkw = ctx->d_fkw; A.fval = (const R*)ctx->d_fval;
    A.nnz = ctx->dict_nnz; A.wts = (const R*)ctx->d_w;
    A.caps = sparse_caps(ctx->W, packed);
    const bool lists = use_row_lists(ctx) && ctx->d_rl_cnt != nullptr;
    A.rl_cnt = lists ? ctx->d_rl_cnt : nullptr; A.rl_f = ctx->d_rl_f; A.rl_cap = kRowListCap; A.rl_filled = ctx->rl_filled ? 1 : 0;
    return A;
}

// More signals than two per CU: the four-workgroups-per-CU form of the loop (see SparseRecorr) when its LDS fits.
static bool sparse_loop_packed(const DevParams& P, size_t lds_packed)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    if (getenv("HSCMP_SPARSE_PACKED")) return atoi(getenv("HSCMP_SPARSE_PACKED")) != 0;
    return P.B > 2 * cus && lds_packed <= (size_t)40 * 1024;
}

template <typename R, bool PACKED> static int launch_iterate_sparse_t(hscmp_ctx* ctx, const DevParams& P0, bool dry, size_t* lds_out)
{
    using Pol = SparseRecorr<R, PACKED>;
    State<R> S = make_state<R>(ctx);
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    const SparseArgs<R> A = sparse_args<R>(ctx, P.T, PACKED);
    const size_t lds = ((sizeof(typename Pol::Shared) + 15) / 16) * 16 + Pol::extra_lds_bytes(P, A);
    if (lds_out) *lds_out = lds;
    if (dry) return HSCMP_OK;
    auto kern = iterate_kernel<R, Pol>;
    HIP_TRY(ctx, set_dyn_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3(P.B), dim3(kThreads), lds, ctx->stream, P, S, A);
    return HSCMP_OK;
}

template <typename R> static int launch_iterate_sparse(hscmp_ctx* ctx, const DevParams& P0)
{
    ctx->rp_last = false;
    if constexpr (sizeof(R) == 8) {
        // small batches of blocked rounds: the round-parallel loop (hscmp_rp_sparse.h), one wave per atom of the round
        if (use_rp(P0, true)) {
            State<R> S = make_state<R>(ctx);
            const SparseArgs<R> A = sparse_args<R>(ctx, P0.T, false);
            if (rp_sparse_launch<R>(ctx->stream, P0, S, A, true) == 0) {
                if (rp_sparse_launch<R>(ctx->stream, P0, S, A, false) != 0) return fail(ctx, HSCMP_ERR_HIP, "the round-parallel level loop could not be launched");
                ctx->rp_last = true;
                return HSCMP_OK;
            }
        }
    }
    size_t lds_packed = 0;
    (void)launch_iterate_sparse_t<R, true>(ctx, P0, true, &lds_packed);
    if (sparse_loop_packed(P0, lds_packed)) return launch_iterate_sparse_t<R, true>(ctx, P0, false, nullptr);
    return launch_iterate_sparse_t<R, false>(ctx, P0, false, nullptr);
}

template <typename R> static int launch_corr_init_sparse(hscmp_ctx* ctx, const DevParams& P)
{
    State<R> S = make_state<R>(ctx);
    const SparseArgs<R> A = sparse_args<R>(ctx, P.T);
    const size_t lds = sparse_lds_bytes<R>(A.caps) + staged_dict_bytes(P, A) + (size_t)((P.T + 31) / 32) * sizeof(unsigned);
    auto kern = corr_init_sparse_kernel<R>;
    HIP_TRY(ctx, set_dyn_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3(P.B, sparse_init_split(P.B, P.T, P.W)), dim3(kThreads), lds, ctx->stream, P, S, A);
    return HSCMP_OK;
}

template <typename R> static int launch_iterate(hscmp_ctx* ctx, const DevParams& P0)
{
    State<R> S = make_state<R>(ctx);
    DevParams P = P0;
    set_segments(P, GenericRecorr<R>::kMaxSegments);
    const size_t lds = ((sizeof(typename GenericRecorr<R>::Shared) + 15) / 16) * 16 + GenericRecorr<R>::extra_lds_bytes(P);
    auto kern = iterate_kernel<R, GenericRecorr<R>>;
    HIP_TRY(ctx, set_dyn_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3(P.B), dim3(kThreads), lds, ctx->stream, P, S, typename GenericRecorr<R>::Args{});
    return HSCMP_OK;
}

// LoCOMP (hscmp_locomp.h): the table-free dense loop with the group re-fit as its atom body
template <typename R> static int launch_iterate_locomp(hscmp_ctx* ctx, const DevParams& P0)
{
    using Pol = LocompRecorr<R>;
    State<R> S = make_state<R>(ctx);
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    const size_t lds = ((sizeof(typename Pol::Shared) + 15) / 16) * 16 + Pol::extra_lds_bytes(P);
    auto kern = iterate_kernel<R, Pol>;
    HIP_TRY(ctx, set_dyn_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3(P.B), dim3(kThreads), lds, ctx->stream, P, S, typename Pol::Args{});
    return HSCMP_OK;
}

// single-feature float32 with a dictionary image (hscmp_set_dictionary built it): the re-correlations on the matrix cores.
// Two signals per workgroup (one image per CU, two signals on its matrix pipe) when the batch has more signals than the chip has CUs.
template <int S4C, bool HAS_W, int GS> static int launch_iterate_locomp_mfma_g(hscmp_ctx* ctx, const DevParams& P0, bool dry)
{
    using Pol = LocompMfma<S4C, HAS_W, GS>;
    State<float> S = make_state<float>(ctx);
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    MfmaArgs A;
    A.dimg = (const float*)ctx->d_Dfrag; A.G = mfma_groups(P.K); A.S4 = S4C; A.has_w = HAS_W ? 1 : 0;
    const size_t lds = Pol::total_lds_bytes(P, A);
    if (lds > (size_t)158 * 1024) return -1;
    if (dry) return 0;
    auto kern = iterate_kernel<float, Pol>;
    HIP_TRY(ctx, set_dyn_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3((P.B + GS - 1) / GS), dim3(GS * kThreads), lds, ctx->stream, P, S, A);
    return HSCMP_OK;
}
template <int S4C, bool HAS_W> static int launch_iterate_locomp_mfma_t(hscmp_ctx* ctx, const DevParams& P, bool dry)
{
    // signals per workgroup: as many as it takes to put the whole batch on the chip at once -- two or four around one dictionary
    // image, their tiles sharing the CU's matrix pipe (HSCMP_LOCOMP_PACK = 1 / 2 / 4 overrides)
    const char* e = getenv("HSCMP_LOCOMP_PACK");
    const int cus = mfma_device_cus();
    const int pack = e ? atoi(e) : P.B > 2 * cus ? 4 : P.B > cus ? 2 : 1;
    if (pack >= 4 && launch_iterate_locomp_mfma_g<S4C, HAS_W, 4>(ctx, P, true) == 0) return launch_iterate_locomp_mfma_g<S4C, HAS_W, 4>(ctx, P, dry);
    if (pack >= 2 && launch_iterate_locomp_mfma_g<S4C, HAS_W, 2>(ctx, P, true) == 0) return launch_iterate_locomp_mfma_g<S4C, HAS_W, 2>(ctx, P, dry);
    return launch_iterate_locomp_mfma_g<S4C, HAS_W, 1>(ctx, P, dry);
}
static int launch_iterate_locomp_mfma(hscmp_ctx* ctx, const DevParams& P, bool dry)
{
    if (ctx->dtype != HSCMP_F32 || ctx->F != 1 || !ctx->d_Dfrag || getenv("HSCMP_LOCOMP_NO_MFMA")) return -1;
    const bool w = ctx->d_w != nullptr;
    switch (mfma_chunks(P.W)) {
    case 8: return w ? launch_iterate_locomp_mfma_t<8, true>(ctx, P, dry) : launch_iterate_locomp_mfma_t<8, false>(ctx, P, dry);
    case 4: return w ? launch_iterate_locomp_mfma_t<4, true>(ctx, P, dry) : launch_iterate_locomp_mfma_t<4, false>(ctx, P, dry);
    case 2: return w ? launch_iterate_locomp_mfma_t<2, true>(ctx, P, dry) : launch_iterate_locomp_mfma_t<2, false>(ctx, P, dry);
    default: return -1;
    }
}

// (dry: only tells whether the policy's LDS fits -- staged dictionary lists can be too long; the dense form runs then)
template <typename R> static int launch_iterate_locomp_sparse(hscmp_ctx* ctx, const DevParams& P0, bool dry = false)
{
    using Pol = LocompSparse<R>;
    State<R> S = make_state<R>(ctx);
    DevParams P = P0;
    set_segments(P, Pol::kMaxSegments);
    const SparseArgs<R> A = sparse_args<R>(ctx, P.T, false);
    const size_t lds = ((sizeof(typename Pol::Shared) + 15) / 16) * 16 + Pol::extra_lds_bytes(P, A);
    if (lds > (size_t)158 * 1024) return -1;
    if (dry) return 0;
    auto kern = iterate_kernel<R, Pol>;
    HIP_TRY(ctx, set_dyn_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3(P.B), dim3(kThreads), lds, ctx->stream, P, S, A);
    return HSCMP_OK;
}

// Slots of the previous level an input was scattered from (level chaining): the input then already sits in the
// residual buffer and prepare only needs the energy.
struct ChainSource { const int* slot_t; const int* slot_k; const double* slot_a; const int* stats; int cap, first, has_min; double minc; bool lists; int max_slots; };

template <typename R> static int run_encode(hscmp_ctx* ctx, const DevParams& P, const void* x_dev, const ChainSource* chain = nullptr)
{
    State<R> S = make_state<R>(ctx);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (chain) {
        // long slot lists: counting sort in LDS when a word per slot fits (cell index / 256 and slot number in 32 bits)
        int ibits = 1;
        while ((1 << ibits) < std::max(2, chain->max_slots)) ++ibits;
        const long long cells256 = ((long long)P.T * P.F + 255) >> 8;
        const size_t lds = (size_t)std::max(1, chain->max_slots) * sizeof(unsigned);
        const int sorted_min = getenv("HSCMP_SORTED_PREPARE_MIN") ? atoi(getenv("HSCMP_SORTED_PREPARE_MIN")) : 2048;
        const bool sorted = chain->max_slots > sorted_min && lds <= (size_t)150 * 1024 && ibits < 31 && cells256 < (1ll << (32 - ibits)) &&
                            !getenv("HSCMP_NO_SORTED_PREPARE") &&
                            set_dyn_lds((const void*)prepare_from_slots_sorted_kernel<R>, lds) == hipSuccess;
        if (sorted)
            hipLaunchKernelGGL((prepare_from_slots_sorted_kernel<R>), dim3(P.B), dim3(kThreads), lds, ctx->stream, P, S, chain->slot_t, chain->slot_k,
                               chain->slot_a, chain->stats, chain->cap, chain->first, chain->has_min, chain->minc, ibits);
        else
            hipLaunchKernelGGL((prepare_from_slots_kernel<R>), dim3(P.B), dim3(kThreads), 0, ctx->stream, P, S, chain->slot_t, chain->slot_k,
                               chain->slot_a, chain->stats, chain->cap, chain->first, chain->has_min, chain->minc,
                               chain->lists ? ctx->d_rl_cnt : nullptr, ctx->d_rl_f, kRowListCap);
    } else
        hipLaunchKernelGGL((prepare_kernel<R>), dim3(P.B), dim3(kThreads), 0, ctx->stream, P, S, (const R*)x_dev);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    // The matrix-core kernels come as a pair: the score-only state the initial correlation leaves is what the MFMA loop
    // reads (the generic / sparse kernels keep coefficient + atom instead).  Both are configured before anything is
    // queued; if either cannot run this shape, neither does.
    bool mf = false;
    const bool loc = ctx->method == HSCMP_METHOD_LOCOMP;       // (its loop keeps coefficient + atom per position: no score-only state)
    if (!loc && use_mfma(ctx, P.T) && mfma_launch_corr_init<R>(ctx->stream, P, S, (const R*)ctx->d_Dfrag, true) == 0 &&
        mfma_launch_iterate<R>(ctx->stream, P, S, (const R*)ctx->d_Dfrag, true) == 0) {
        if (mfma_launch_corr_init<R>(ctx->stream, P, S, (const R*)ctx->d_Dfrag) != 0)
            return fail(ctx, HSCMP_ERR_HIP, "the MFMA initial correlation could not be launched");
        mf = true;
    }
    ctx->mfma_state = mf;
    if (!mf && use_sparse_loop(ctx) && use_row_lists(ctx) && !ctx->rl_filled) {
        // per-row lists of the input's non-zero cells (the level chaining writes them while it scatters)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_rl_cnt, 0, (size_t)P.B * P.T * sizeof(int), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_rl_f, 0xff, (size_t)P.B * P.T * kRowListCap * sizeof(int), ctx->stream));
        const int split = std::max(1, std::min(256, 4096 / P.B));
        hipLaunchKernelGGL((build_row_lists_kernel<R>), dim3(P.B, split), dim3(kThreads), 0, ctx->stream, (const R*)x_dev, P.T, P.F,
                           ctx->d_rl_cnt, ctx->d_rl_f, kRowListCap);
    }
    const bool spi = !mf && use_sparse_init(ctx, P.T);
    // (LoCOMP on the matrix cores: its loop kernel starts with the initial correlation -- see LocompMfma::prologue)
    const bool own_init = loc && !spi && !use_sparse_loop(ctx) && launch_iterate_locomp_mfma(ctx, P, true) == 0;
    if (spi) { int rc = launch_corr_init_sparse<R>(ctx, P); if (rc) return rc; }
    if (!mf && !spi && !own_init) {
        dim3 grid((P.T + kThreads - 1) / kThreads, P.B);
        hipLaunchKernelGGL((corr_init_generic_kernel<R, false>), grid, dim3(kThreads), 0, ctx->stream, P, S,
                           (const R*)ctx->d_resid, P.off, P.T, (R*)nullptr);
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    bool mfi = false;
    if (mf) {
        if (launch_mfma_loop<R>(ctx, P, S) != 0)
            return fail(ctx, HSCMP_ERR_HIP, "the MFMA loop could not be launched on the state of the MFMA initial correlation");
        mfi = true;
    }
    const bool spl = !mfi && !loc && use_sparse_loop(ctx);
    ctx->loop_kept_lists = spl && use_row_lists(ctx);
    ctx->locomp_state = loc;
    const bool locs = loc && use_sparse_loop(ctx) && launch_iterate_locomp_sparse<R>(ctx, P, true) == 0;
    ctx->locomp_sparse = locs;
    if (loc) ctx->loop_kept_lists = locs && use_row_lists(ctx);
    const bool locm = loc && !locs && own_init;
    ctx->locomp_mfma = locm;
    if (loc) {
        ctx->rp_last = false;
        int rc = locs ? launch_iterate_locomp_sparse<R>(ctx, P) : locm ? launch_iterate_locomp_mfma(ctx, P, false) : launch_iterate_locomp<R>(ctx, P);
        if (rc) return rc;
    }
    else if (spl) { int rc = launch_iterate_sparse<R>(ctx, P); if (rc) return rc; }
    else if (!mfi) { int rc = launch_iterate<R>(ctx, P); if (rc) return rc; }
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    ctx->timed = true; ctx->timed_loop_only = false;
    ctx->variant = std::string(mf ? "mfma" : own_init ? "own" : spi ? (ctx->d_nzptr ? "dictlist" : "sparse") : "generic") + "_init+" + (loc ? (locs ? "locomp_dictlist" : locm ? "locomp_mfma" : "locomp") : mfi ? "mfma" : spl ? (ctx->d_nzptr ? "dictlist" : "gathered") : "generic") +
                   "_loop_" + (sizeof(R) == 4 ? "f32" : "f64") + ((mfi || spl) && ctx->rp_last ? std::string("_rp") : mfi && mfma_last_group() > 1 ? "_x" + std::to_string(mfma_last_group()) : std::string());
    return HSCMP_OK;
}

static int encode_common(hscmp_ctx* ctx, const void* x, bool host, int B, int T, const hscmp_params* params)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_encode_batch: ctx is NULL");
    if (ctx->dtype < 0) return fail(ctx, HSCMP_ERR_STATE, "hscmp_encode_batch: no dictionary set");
    if (!x || !params || B <= 0 || T <= 0) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_encode_batch: bad arguments (B=%d T=%d)", B, T);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevParams P;
    int rc = make_params(ctx, B, T, params, &P);
    if (rc) return rc;
    if ((rc = ensure_workspace(ctx, P, host))) return rc;
    const void* xd = x;
    if (host) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_x, x, (size_t)B * T * ctx->F * esize(ctx->dtype), hipMemcpyHostToDevice, ctx->stream));
        xd = ctx->d_x;
    }
    ctx->P = P; ctx->last = *params; ctx->B = B; ctx->T = T; ctx->cap = P.cap; ctx->maxsel = P.maxsel;
    ctx->listed_rows = 0;                       // the residual buffer is overwritten with a dense input
    ctx->last_x_dev = xd;
    rc = ctx->dtype == HSCMP_F32 ? run_encode<float>(ctx, P, xd) : run_encode<double>(ctx, P, xd);
    if (rc) return rc;
    ctx->have_batch = true;
    if (host) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_encode_batch(hscmp_ctx* ctx, const void* x, int B, int T, const hscmp_params* params)
{
    return encode_common(ctx, x, true, B, T, params);
}

extern "C" int hscmp_encode_batch_device(hscmp_ctx* ctx, const void* x_dev, int B, int T, const hscmp_params* params)
{
    return encode_common(ctx, x_dev, false, B, T, params);
}

extern "C" int hscmp_encode_batch_from_level(hscmp_ctx* ctx, hscmp_ctx* prev, int first, int count, double min_coefficients,
                                             const hscmp_params* params)
{
    if (!ctx || !prev) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_encode_batch_from_level: NULL context");
    if (ctx->dtype != HSCMP_F64) return fail(ctx, HSCMP_ERR_STATE, "hscmp_encode_batch_from_level: the level dictionary must be float64");
    if (!prev->have_batch) return fail(ctx, HSCMP_ERR_STATE, "hscmp_encode_batch_from_level: the previous level has no results");
    if (ctx->device != prev->device) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_encode_batch_from_level: contexts on different GPUs");
    if (ctx->F != prev->K) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_encode_batch_from_level: F=%d of this level != K=%d of the previous one", ctx->F, prev->K);
    if (!params || first < 0 || count <= 0 || first + count > prev->B) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_encode_batch_from_level: bad signal range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(prev->stream));           // the previous level's results are final
    const int T = prev->T;
    DevParams P;
    int rc = make_params(ctx, count, T, params, &P);
    if (rc) return rc;
    if ((rc = ensure_workspace(ctx, P, false))) return rc;        // no input buffer: the slots are scattered straight into the residual
    const size_t bytes = (size_t)count * T * ctx->F * sizeof(double);
    const bool lists = use_sparse_loop(ctx) && use_row_lists(ctx);
    if (ctx->listed_rows > 0 && ctx->listed_F == ctx->F && !getenv("HSCMP_NO_LAZY_CLEAR")) {
        // the buffer still holds the previous chained batch; its lists say where
        hipLaunchKernelGGL((clear_listed_cells_kernel<double>), dim3((unsigned)((ctx->listed_rows + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                           (double*)ctx->d_resid, ctx->listed_rows, ctx->F, ctx->d_rl_cnt, ctx->d_rl_f, kRowListCap);
        const size_t covered = (size_t)ctx->listed_rows * ctx->F * sizeof(double);
        if (bytes > covered) HIP_TRY(ctx, hipMemsetAsync((char*)ctx->d_resid + covered, 0, bytes - covered, ctx->stream));
    } else {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_resid, 0, bytes, ctx->stream));
    }
    ctx->listed_rows = 0;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_rowflag, 0, (size_t)count * T, ctx->stream));
    if (lists) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_rl_cnt, 0, (size_t)count * T * sizeof(int), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_rl_f, 0xff, (size_t)count * T * kRowListCap * sizeof(int), ctx->stream));
    }
    const int has_min = !std::isnan(min_coefficients);
    hipLaunchKernelGGL((scatter_slots_kernel<double>), dim3(count), dim3(kThreads), 0, ctx->stream, (double*)ctx->d_resid, T, ctx->F,
                       prev->d_slot_t, prev->d_slot_k, prev->d_slot_a, prev->d_stats, prev->cap, first, has_min,
                       has_min ? min_coefficients : 0.0, ctx->d_rowflag, lists ? ctx->d_rl_cnt : nullptr, ctx->d_rl_f, kRowListCap);
    ctx->P = P; ctx->last = *params; ctx->B = count; ctx->T = T; ctx->cap = P.cap; ctx->maxsel = P.maxsel;
    ctx->rowflag_valid = true;                  // the sparse initial correlation skips its scan of the dense input
    ctx->rl_filled = lists;
    // (the longest slot list of the range sizes the LDS of the energy kernel: the previous level's counters are final)
    int max_slots = 0;
    {
        std::vector<int> pst((size_t)count * ST_COUNT);
        HIP_TRY(ctx, hipMemcpy(pst.data(), prev->d_stats + (size_t)first * ST_COUNT, pst.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < count; ++i) max_slots = std::max(max_slots, pst[(size_t)i * ST_COUNT + ST_SLOTS]);
    }
    const ChainSource chain{prev->d_slot_t, prev->d_slot_k, prev->d_slot_a, prev->d_stats, prev->cap, first, has_min,
                            has_min ? min_coefficients : 0.0, lists, max_slots};
    rc = run_encode<double>(ctx, P, ctx->d_resid, &chain);
    ctx->rowflag_valid = false; ctx->rl_filled = false;
    if (rc) return rc;
    ctx->have_batch = true;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // (only now: a failed launch leaves the buffer in an unknown state, and so does any other writer -- see the resets)
    if (lists && ctx->loop_kept_lists) { ctx->listed_rows = (int64_t)count * T; ctx->listed_F = ctx->F; }
    return HSCMP_OK;
}

extern "C" int hscmp_continue(hscmp_ctx* ctx, int max_rounds)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_continue: ctx is NULL");
    if (!ctx->have_batch) return fail(ctx, HSCMP_ERR_STATE, "hscmp_continue: no batch encoded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevParams P = ctx->P;
    P.max_rounds = max_rounds;
    bool mfi = false;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));       // hscmp_last_kernel_ms: [2] = this launch, [0] = [1] = 0
    ctx->timed_loop_only = true;
    if (ctx->mfma_state) {          // the loop must be the one that understands the state the encode left
        int rc;
        if (ctx->dtype == HSCMP_F32) {
            State<float> S = make_state<float>(ctx);
            rc = launch_mfma_loop<float>(ctx, P, S);
        } else {
            State<double> S = make_state<double>(ctx);
            rc = launch_mfma_loop<double>(ctx, P, S);
        }
        if (rc != 0) return fail(ctx, HSCMP_ERR_HIP, "hscmp_continue: the MFMA loop could not be launched");
        mfi = true;
    }
    if (ctx->locomp_state) {
        int rc = ctx->locomp_mfma ? launch_iterate_locomp_mfma(ctx, P, false) :
                 ctx->locomp_sparse ? (ctx->dtype == HSCMP_F32 ? launch_iterate_locomp_sparse<float>(ctx, P) : launch_iterate_locomp_sparse<double>(ctx, P))
                                    : (ctx->dtype == HSCMP_F32 ? launch_iterate_locomp<float>(ctx, P) : launch_iterate_locomp<double>(ctx, P));
        if (rc) return rc;
    } else if (!mfi && use_sparse_loop(ctx)) {
        int rc = ctx->dtype == HSCMP_F32 ? launch_iterate_sparse<float>(ctx, P) : launch_iterate_sparse<double>(ctx, P);
        if (rc) return rc;
    } else if (!mfi) {
        int rc = ctx->dtype == HSCMP_F32 ? launch_iterate<float>(ctx, P) : launch_iterate<double>(ctx, P);
        if (rc) return rc;
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_grow_events(hscmp_ctx* ctx, int new_max_events)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_grow_events: ctx is NULL");
    if (!ctx->have_batch) return fail(ctx, HSCMP_ERR_STATE, "hscmp_grow_events: no batch encoded");
    if (new_max_events <= ctx->cap) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_grow_events: %d is not above the current capacity %d", new_max_events, ctx->cap);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t es = esize(ctx->dtype), B = (size_t)ctx->B, oc = (size_t)ctx->cap, nc = (size_t)new_max_events;
    struct Row { void** p; size_t* cap; size_t elem; void* fresh; };
    Row rows[] = {{(void**)&ctx->d_ev_t, &ctx->caps[4], 4, nullptr}, {(void**)&ctx->d_ev_k, &ctx->caps[5], 4, nullptr}, {(void**)&ctx->d_ev_c, &ctx->caps[6], es, nullptr},
                  {(void**)&ctx->d_slot_t, &ctx->caps[7], 4, nullptr}, {(void**)&ctx->d_slot_k, &ctx->caps[8], 4, nullptr}, {(void**)&ctx->d_slot_a, &ctx->caps[9], 8, nullptr}};
    // the slot hash table follows the capacity; the loop rebuilds its contents from the slot list on the next launch
    const unsigned hmask = slot_hash_mask(new_max_events);
    struct Tab { void** p; size_t* cap; size_t elem; void* fresh; };
    Tab tabs[] = {{(void**)&ctx->d_hkey, &ctx->cap_hkey, sizeof(unsigned long long), nullptr}, {(void**)&ctx->d_hval, &ctx->cap_hval, sizeof(int), nullptr}};
    // every new buffer first; the context changes only when all of them exist (a failed call leaves the batch as it was)
    hipError_t e = hipSuccess;
    size_t failed_bytes = 0;
    for (Row& r : rows) {
        if ((e = hipMalloc(&r.fresh, B * nc * r.elem)) != hipSuccess) { failed_bytes = B * nc * r.elem; break; }
    }
    if (e == hipSuccess)
        for (Tab& t : tabs) {
            const size_t bytes = B * ((size_t)hmask + 1) * t.elem;
            if (*t.p && *t.cap >= bytes) continue;
            if ((e = hipMalloc(&t.fresh, bytes)) != hipSuccess) { failed_bytes = bytes; break; }
        }
    if (e == hipSuccess)
        for (Row& r : rows)
            if ((e = hipMemcpy2D(r.fresh, nc * r.elem, *r.p, oc * r.elem, oc * r.elem, B, hipMemcpyDeviceToDevice)) != hipSuccess) break;
    std::vector<int> stats(B * ST_COUNT);
    if (e == hipSuccess) e = hipMemcpy(stats.data(), ctx->d_stats, stats.size() * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        for (size_t b = 0; b < B; ++b)
            if (stats[b * ST_COUNT + ST_STOP] == STOP_CAPACITY) stats[b * ST_COUNT + ST_STOP] = STOP_RUNNING;
        e = hipMemcpy(ctx->d_stats, stats.data(), stats.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        for (Row& r : rows) if (r.fresh) (void)hipFree(r.fresh);
        for (Tab& t : tabs) if (t.fresh) (void)hipFree(t.fresh);
        return fail(ctx, failed_bytes ? HSCMP_ERR_ALLOC : HSCMP_ERR_HIP, "hscmp_grow_events: %s (%zu bytes)", hipGetErrorString(e), failed_bytes);
    }
    for (Row& r : rows) { (void)hipFree(*r.p); *r.p = r.fresh; *r.cap = B * nc * r.elem; }
    for (Tab& t : tabs) if (t.fresh) { if (*t.p) (void)hipFree(*t.p); *t.p = t.fresh; *t.cap = B * ((size_t)hmask + 1) * t.elem; }
    ctx->cap = new_max_events; ctx->P.cap = new_max_events; ctx->P.hmask = hmask; ctx->last.max_events = new_max_events;
    return HSCMP_OK;
}

extern "C" int hscmp_stop_signal(hscmp_ctx* ctx, int b)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_stop_signal: ctx is NULL");
    if (!ctx->have_batch || b < 0 || b >= ctx->B) return fail(ctx, HSCMP_ERR_STATE, "hscmp_stop_signal: bad signal index %d", b);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int v = STOP_CALLBACK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int cur = 0;
    HIP_TRY(ctx, hipMemcpy(&cur, ctx->d_stats + (size_t)b * ST_COUNT + ST_STOP, sizeof(int), hipMemcpyDeviceToHost));
    if (cur == STOP_RUNNING)
        HIP_TRY(ctx, hipMemcpy(ctx->d_stats + (size_t)b * ST_COUNT + ST_STOP, &v, sizeof(int), hipMemcpyHostToDevice));
    return HSCMP_OK;
}

static int fetch(hscmp_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!dst) return HSCMP_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return HSCMP_OK;
}

#define NEED_BATCH(ctx, name)                                                                   \
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, name ": ctx is NULL");                    \
    if (!ctx->have_batch) return fail(ctx, HSCMP_ERR_STATE, name ": no batch encoded");         \
    HIP_TRY(ctx, hipSetDevice(ctx->device));

extern "C" int hscmp_fetch_events(hscmp_ctx* ctx, int32_t* ev_t, int32_t* ev_k, void* ev_c)
{
    NEED_BATCH(ctx, "hscmp_fetch_events");
    const size_t n = (size_t)ctx->B * ctx->cap;
    int rc;
    if ((rc = fetch(ctx, ev_t, ctx->d_ev_t, n * 4))) return rc;
    if ((rc = fetch(ctx, ev_k, ctx->d_ev_k, n * 4))) return rc;
    if ((rc = fetch(ctx, ev_c, ctx->d_ev_c, n * esize(ctx->dtype)))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_fetch_slots(hscmp_ctx* ctx, int32_t* slot_t, int32_t* slot_k, double* slot_acc)
{
    NEED_BATCH(ctx, "hscmp_fetch_slots");
    const size_t n = (size_t)ctx->B * ctx->cap;
    int rc;
    if ((rc = fetch(ctx, slot_t, ctx->d_slot_t, n * 4))) return rc;
    if ((rc = fetch(ctx, slot_k, ctx->d_slot_k, n * 4))) return rc;
    if ((rc = fetch(ctx, slot_acc, ctx->d_slot_a, n * 8))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_fetch_stats(hscmp_ctx* ctx, int32_t* stats)
{
    NEED_BATCH(ctx, "hscmp_fetch_stats");
    int rc = fetch(ctx, stats, ctx->d_stats, (size_t)ctx->B * ST_COUNT * sizeof(int));
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_fetch_residual(hscmp_ctx* ctx, void* residual)
{
    NEED_BATCH(ctx, "hscmp_fetch_residual");
    int rc = fetch(ctx, residual, ctx->d_resid, (size_t)ctx->B * ctx->T * ctx->F * esize(ctx->dtype));
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_fetch_energies(hscmp_ctx* ctx, double* energies)
{
    NEED_BATCH(ctx, "hscmp_fetch_energies");
    if (!energies) return HSCMP_OK;
    const size_t n = (size_t)ctx->B * 2;
    if (ctx->dtype == HSCMP_F64) {
        int rc = fetch(ctx, energies, ctx->d_energy, n * 8);
        if (rc) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        std::vector<float> tmp(n);
        int rc = fetch(ctx, tmp.data(), ctx->d_energy, n * 4);
        if (rc) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < n; ++i) energies[i] = (double)tmp[i];
    }
    return HSCMP_OK;
}

extern "C" int hscmp_get_device_view(hscmp_ctx* ctx, hscmp_device_view* v)
{
    NEED_BATCH(ctx, "hscmp_get_device_view");
    if (!v) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_get_device_view: view is NULL");
    v->B = ctx->B; v->T = ctx->T; v->F = ctx->F; v->K = ctx->K; v->W = ctx->W; v->max_events = ctx->cap;
    v->dtype = ctx->dtype; v->reserved = 0;
    v->ev_t = ctx->d_ev_t; v->ev_k = ctx->d_ev_k; v->ev_c = ctx->d_ev_c; v->stats = ctx->d_stats;
    v->residual = ctx->d_resid; v->energies = ctx->d_energy; v->best_c = ctx->d_best_c; v->best_k = ctx->d_best_k;
    return HSCMP_OK;
}

extern "C" int hscmp_last_kernel_ms(hscmp_ctx* ctx, float* out4)
{
    if (!ctx || !out4) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_last_kernel_ms: NULL argument");
    if (!ctx->timed) return fail(ctx, HSCMP_ERR_STATE, "hscmp_last_kernel_ms: nothing timed yet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[3]));
    out4[0] = out4[1] = out4[3] = 0.f;
    for (int i = ctx->timed_loop_only ? 2 : 0; i < 3; ++i) HIP_TRY(ctx, hipEventElapsedTime(&out4[i], ctx->ev[i], ctx->ev[i + 1]));
    return HSCMP_OK;
}

extern "C" int hscmp_mem_info(hscmp_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes)
{
    if (!ctx || !free_bytes || !total_bytes) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_mem_info: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    HIP_TRY(ctx, hipMemGetInfo(&f, &t));
    *free_bytes = f; *total_bytes = t;
    return HSCMP_OK;
}

extern "C" const char* hscmp_last_variant(hscmp_ctx* ctx) { return ctx ? ctx->variant.c_str() : ""; }

extern "C" int hscmp_copy_from_device(hscmp_ctx* ctx, const void* src_dev, uint64_t nbytes, void* dst_host)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_copy_from_device: ctx is NULL");
    if (!src_dev || !dst_host) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_copy_from_device: NULL argument");
    if (nbytes == 0) return HSCMP_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

// modeling.py:149-188 convolve1d on the GPU: full table out [Tout][K]
template <typename R> static void launch_convolve(hscmp_ctx* ctx, const R* dx, int T, int same, int Tout, R* dout)
{
    DevParams P{};
    P.B = 1; P.T = T; P.K = ctx->K; P.W = ctx->W; P.F = ctx->F; P.off = (ctx->W - 1) / 2;
    State<R> S{};
    S.D = (const R*)ctx->d_D; S.weights = nullptr;
    S.Dc = ctx->d_Dc ? (const R*)ctx->d_Dc : (const R*)ctx->d_D;
    dim3 grid((Tout + kThreads - 1) / kThreads, 1);
    hipLaunchKernelGGL((corr_init_generic_kernel<R, true>), grid, dim3(kThreads), 0, ctx->stream, P, S, dx, same ? P.off : 0, Tout, dout);
}

template <typename R> static int run_convolve(hscmp_ctx* ctx, const void* x, int T, int same, void* out)
{
    const int K = ctx->K, W = ctx->W, F = ctx->F;
    const int Tout = same ? T : T - W + 1;
    if (Tout <= 0) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_convolve1d: T=%d shorter than the filters (W=%d)", T, W);
    int rc;
    if ((rc = epi_buffer(ctx, kArenaRowA, (size_t)T * F * sizeof(R))) != HSCMP_OK) return rc;
    if ((rc = epi_buffer(ctx, kArenaRowB, (size_t)Tout * K * sizeof(R))) != HSCMP_OK) return rc;
    R* dx = (R*)ctx->d_epi[kArenaRowA]; R* dout = (R*)ctx->d_epi[kArenaRowB];
    HIP_TRY(ctx, hipMemcpyAsync(dx, x, (size_t)T * F * sizeof(R), hipMemcpyHostToDevice, ctx->stream));
    launch_convolve<R>(ctx, dx, T, same, Tout, dout);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, dout, (size_t)Tout * K * sizeof(R), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_convolve1d(hscmp_ctx* ctx, const void* x, int T, int same, void* out)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_convolve1d: ctx is NULL");
    if (ctx->dtype < 0) return fail(ctx, HSCMP_ERR_STATE, "hscmp_convolve1d: no dictionary set");
    if (!x || !out || T <= 0) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_convolve1d: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ctx->dtype == HSCMP_F32 ? run_convolve<float>(ctx, x, T, same, out) : run_convolve<double>(ctx, x, T, same, out);
}

// modeling.py:454-460: per-window best (position, atom, coefficient) of the k-means learner
template <typename R>
static int run_assign(hscmp_ctx* ctx, const void* windows, int N, int L, int32_t* out_t, int32_t* out_k, void* out_c)
{
    const int K = ctx->K, W = ctx->W, F = ctx->F;
    const size_t wbytes = (size_t)N * L * F * sizeof(R);
    int rc;
    if ((rc = epi_buffer(ctx, kArenaRowA, wbytes)) != HSCMP_OK) return rc;
    if ((rc = epi_buffer(ctx, kArenaRowB, (size_t)N * sizeof(int))) != HSCMP_OK) return rc;
    if ((rc = epi_buffer(ctx, kArenaRowC, (size_t)N * sizeof(int))) != HSCMP_OK) return rc;
    if ((rc = epi_buffer(ctx, kArenaRowD, (size_t)N * sizeof(R))) != HSCMP_OK) return rc;
    R* dwin = (R*)ctx->d_epi[kArenaRowA]; int* dt = (int*)ctx->d_epi[kArenaRowB]; int* dk = (int*)ctx->d_epi[kArenaRowC]; R* dc = (R*)ctx->d_epi[kArenaRowD];
    HIP_TRY(ctx, hipMemcpyAsync(dwin, windows, wbytes, hipMemcpyHostToDevice, ctx->stream));
    const int lds_elems = (size_t)L * F * sizeof(R) <= 32768 ? L * F : 0;
    hipLaunchKernelGGL((assign_windows_kernel<R>), dim3(N), dim3(kThreads), (size_t)lds_elems * sizeof(R), ctx->stream,
                       (const R*)dwin, L, K, W, F, (const R*)ctx->d_D, lds_elems, dt, dk, dc);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_t, dt, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_k, dk, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (out_c) HIP_TRY(ctx, hipMemcpyAsync(out_c, dc, (size_t)N * sizeof(R), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_assign_windows(hscmp_ctx* ctx, const void* windows, int N, int L, int32_t* out_t, int32_t* out_k, void* out_c)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_assign_windows: ctx is NULL");
    if (ctx->dtype < 0) return fail(ctx, HSCMP_ERR_STATE, "hscmp_assign_windows: no dictionary set");
    if (!windows || !out_t || !out_k || N <= 0) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_assign_windows: bad arguments");
    if (L < ctx->W) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_assign_windows: windows of %d samples are shorter than the filters (W=%d)", L, ctx->W);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ctx->dtype == HSCMP_F32 ? run_assign<float>(ctx, windows, N, L, out_t, out_k, out_c)
                                   : run_assign<double>(ctx, windows, N, L, out_t, out_k, out_c);
}

// ---- host-side synthesis (modeling.py:226-263 reconstructSignal, sparse branch) ------------------------------
// signal[t - (W-1)/2 + w][f] += c * D[k][w][f] for every event (t, k, c), in the order given, clipped at the borders
// (utils.py:103-131): the reference's sequential overlap-add, float64 accumulation.  Plain host code (no GPU work):
// it is the epilogue of the hierarchical encoder (modeling.py:1596-1611) and releases the Python interpreter lock
// while it runs, so a batch can be post-processed on all cores.
extern "C" int hscmp_host_overlap_add(double* signal, int64_t T, int Fd, const int64_t* rows, const int64_t* cols, const double* data,
                                      int64_t n, const void* D, int W, int dict_is_f32)
{
    if (!signal || !D || (n > 0 && (!rows || !cols || !data)) || T <= 0 || Fd <= 0 || W <= 0) return HSCMP_ERR_INVALID;
    const int64_t lead = (W - 1) / 2, atom = (int64_t)W * Fd;
    for (int64_t i = 0; i < n; ++i) {
        const double c = data[i];
        if (c == 0.0) continue;
        const int64_t p0 = rows[i] - lead;
        const int64_t w0 = p0 < 0 ? -p0 : 0, w1 = p0 + W > T ? T - p0 : W;
        double* dst = signal + (p0 + w0) * Fd;
        if (dict_is_f32) {
            const float* src = (const float*)D + cols[i] * atom + w0 * Fd;
            for (int64_t e = 0; e < (w1 - w0) * Fd; ++e) dst[e] += c * (double)src[e];
        } else {
            const double* src = (const double*)D + cols[i] * atom + w0 * Fd;
            for (int64_t e = 0; e < (w1 - w0) * Fd; ++e) dst[e] += c * src[e];
        }
    }
    return HSCMP_OK;
}

// Host-side CSC assembly of one signal's coefficient slots (modeling.py:1171-1181): drop zeros and |a| < min_coefficients
// (NaN: no clip), order by (atom, position).  indptr [K+1], indices / data [n] (first indptr[K] entries used).
// Slots are distinct (t, k) pairs, so there is nothing to sum.  Plain host code.
extern "C" int hscmp_host_slots_to_csc(const int32_t* slot_t, const int32_t* slot_k, const double* slot_a, int64_t n, int K,
                                       double min_coefficients, int32_t* indptr, int32_t* indices, double* data)
{
    if (n < 0 || K <= 0 || !indptr || (n > 0 && (!slot_t || !slot_k || !slot_a || !indices || !data))) return HSCMP_ERR_INVALID;
    const bool clip = !std::isnan(min_coefficients);
    std::vector<int64_t> keep;
    keep.reserve((size_t)n);
    for (int k = 0; k <= K; ++k) indptr[k] = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double a = slot_a[i];
        if (a == 0.0 || (clip && !(std::fabs(a) >= min_coefficients))) continue;
        if (slot_k[i] < 0 || slot_k[i] >= K) return HSCMP_ERR_INVALID;
        keep.push_back(i);
        indptr[slot_k[i] + 1] += 1;
    }
    for (int k = 0; k < K; ++k) indptr[k + 1] += indptr[k];
    std::vector<int32_t> cur(indptr, indptr + K);
    for (int64_t i : keep) {                       // counting sort by atom ...
        const int32_t o = cur[slot_k[i]]++;
        indices[o] = slot_t[i]; data[o] = slot_a[i];
    }
    for (int k = 0; k < K; ++k) {                  // ... then by position inside each column (short runs: insertion sort)
        const int32_t b0 = indptr[k], b1 = indptr[k + 1];
        if (b1 - b0 > 64) {
            std::vector<std::pair<int32_t, double>> col((size_t)(b1 - b0));
            for (int32_t j = b0; j < b1; ++j) col[(size_t)(j - b0)] = {indices[j], data[j]};
            std::sort(col.begin(), col.end(), [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) { return x.first < y.first; });
            for (int32_t j = b0; j < b1; ++j) { indices[j] = col[(size_t)(j - b0)].first; data[j] = col[(size_t)(j - b0)].second; }
        } else {
            for (int32_t j = b0 + 1; j < b1; ++j) {
                const int32_t ti = indices[j]; const double av = data[j];
                int32_t q = j;
                while (q > b0 && indices[q - 1] > ti) { indices[q] = indices[q - 1]; data[q] = data[q - 1]; --q; }
                indices[q] = ti; data[q] = av;
            }
        }
    }
    return HSCMP_OK;
}

// ---- epilogue of the hierarchical encoder on the device (hscmp_epilogue.h) -----------------------------------------

extern "C" int hscmp_hierarchy_epilogue(hscmp_ctx* last, hscmp_ctx* level0, int first, const hscmp_epilogue_level* levels, int nlevels,
                                        double min_coefficients, const int64_t* offsets, int32_t* out_n, int32_t* out_colptr,
                                        int32_t* out_indices, double* out_data, void* out_events, double* out_residual,
                                        double* out_residual_energy)
{
    if (!last || !level0) return fail(last, HSCMP_ERR_INVALID, "hscmp_hierarchy_epilogue: NULL context");
    if (!last->have_batch || !level0->have_batch) return fail(last, HSCMP_ERR_STATE, "hscmp_hierarchy_epilogue: no batch encoded");
    if (last->device != level0->device) return fail(last, HSCMP_ERR_INVALID, "hscmp_hierarchy_epilogue: contexts on different GPUs");
    if (!levels || nlevels < 1 || nlevels > kEpiMaxLevels || !offsets || !out_n || !out_colptr || !out_indices || !out_data)
        return fail(last, HSCMP_ERR_INVALID, "hscmp_hierarchy_epilogue: bad arguments");
    const int count = last->B, T = last->T, Fd = level0->F, Ktot = last->K;
    if (first < 0 || first + count > level0->B || level0->T != T) return fail(last, HSCMP_ERR_INVALID, "hscmp_hierarchy_epilogue: signal range / length mismatch");
    if (!level0->last_x_dev) return fail(last, HSCMP_ERR_STATE, "hscmp_hierarchy_epilogue: the level-0 input is not on the device any more");
    if (Ktot >= (1 << kEpiColBits) || T >= (1 << kEpiTBits) || last->cap >= (1 << kEpiIdxBits))
        return fail(last, HSCMP_ERR_UNSUPPORTED, "hscmp_hierarchy_epilogue: shape outside the key layout (K < 2^20, T < 2^24, list < 2^20)");
    HIP_TRY(last, hipSetDevice(last->device));
    HIP_TRY(last, hipStreamSynchronize(level0->stream));
    HIP_TRY(last, hipStreamSynchronize(last->stream));
    const long long total = offsets[count];
    EpiArgs A{};
    A.nlevels = nlevels; A.Ktot = Ktot; A.T = T; A.Fd = Fd;
    A.has_min = !std::isnan(min_coefficients); A.minc = A.has_min ? min_coefficients : 0.0;
    // representations: one device buffer, level after level
    size_t rep_bytes = 0;
    for (int l = 0; l < nlevels; ++l) {
        const hscmp_epilogue_level& d = levels[l];
        if (d.col1 > d.col0 && (!d.rep || d.scale <= 0 || d.col1 > Ktot || d.col0 < 0)) return fail(last, HSCMP_ERR_INVALID, "hscmp_hierarchy_epilogue: bad level %d", l);
        rep_bytes += ((size_t)std::max(0, d.col1) * std::max(0, d.scale) * Fd * (d.rep_is_f32 ? 4 : 8) + 15) / 16 * 16;
    }
    int rc;
    if ((rc = epi_buffer(last, 0, std::max<size_t>(16, rep_bytes)))) return rc;
    size_t ro = 0;
    for (int l = 0; l < nlevels; ++l) {
        const hscmp_epilogue_level& d = levels[l];
        EpiLevel& L = A.lv[l];
        L.col0 = d.col0; L.col1 = d.col1; L.scale = d.scale; L.lead = (d.scale - 1) / 2; L.rep_f32 = d.rep_is_f32; L.rep = nullptr;
        if (d.col1 <= d.col0) continue;
        const size_t nb = (size_t)d.col1 * d.scale * Fd * (d.rep_is_f32 ? 4 : 8);       // rows [0, col1): indexed by the column number
        L.rep = (char*)last->d_epi[0] + ro;
        HIP_TRY(last, hipMemcpyAsync((char*)last->d_epi[0] + ro, d.rep, nb, hipMemcpyHostToDevice, last->stream));
        ro += (nb + 15) / 16 * 16;
        A.max_back = std::max(A.max_back, d.scale - 1 - L.lead);
        A.max_fwd = std::max(A.max_fwd, L.lead);
    }
    int nmax = 2;
    while (nmax < last->cap) nmax <<= 1;
    int lds_keys = kEpiLdsKeys;                                  // (HSCMP_EPI_LDS_KEYS: tests force the chunked sort at small sizes)
    if (const char* e = getenv("HSCMP_EPI_LDS_KEYS")) { const int v = atoi(e); if (v >= 64 && v <= kEpiLdsKeys && (v & (v - 1)) == 0) lds_keys = v; }
    const bool need_scratch = true;                              // (the t-sorted keys move there while the LDS holds residual tiles)
    const size_t nres = out_residual ? (size_t)count * T * Fd * sizeof(double) : 0;
    const size_t sizes[8] = {0, (size_t)(count + 1) * sizeof(long long), (size_t)count * sizeof(int), (size_t)count * (Ktot + 1) * sizeof(int),
                             std::max<size_t>(16, (size_t)total * sizeof(int)), std::max<size_t>(16, (size_t)total * sizeof(double)),
                             std::max<size_t>(16, (out_events ? (size_t)total * 16 : 0) + nres),
                             need_scratch ? (size_t)count * nmax * sizeof(unsigned long long) : 0};
    for (int i = 1; i < 8; ++i) if (sizes[i] && (rc = epi_buffer(last, i, sizes[i]))) return rc;
    HIP_TRY(last, hipMemcpyAsync(last->d_epi[1], offsets, sizes[1], hipMemcpyHostToDevice, last->stream));
    A.slot_t = last->d_slot_t; A.slot_k = last->d_slot_k; A.slot_a = last->d_slot_a; A.stats = last->d_stats; A.cap = last->cap;
    A.offsets = (const long long*)last->d_epi[1]; A.out_n = (int*)last->d_epi[2]; A.out_colptr = (int*)last->d_epi[3];
    A.out_indices = (int*)last->d_epi[4]; A.out_data = (double*)last->d_epi[5];
    A.out_events = out_events ? (int*)last->d_epi[6] : nullptr;
    A.out_residual = out_residual ? (double*)((char*)last->d_epi[6] + (out_events ? ((size_t)total * 16 + 15) / 16 * 16 : 0)) : nullptr;
    if (out_events && out_residual && (rc = epi_buffer(last, 6, ((size_t)total * 16 + 15) / 16 * 16 + nres))) return rc;
    A.out_events = out_events ? (int*)last->d_epi[6] : nullptr;
    A.out_residual = out_residual ? (double*)((char*)last->d_epi[6] + (out_events ? ((size_t)total * 16 + 15) / 16 * 16 : 0)) : nullptr;
    A.scratch = (unsigned long long*)last->d_epi[7]; A.scratch_n = nmax;
    A.out_energy = nullptr;
    if (out_residual_energy) {
        if ((rc = epi_buffer(last, kArenaEpiEnergy, (size_t)count * sizeof(double)))) return rc;
        A.out_energy = (double*)last->d_epi[kArenaEpiEnergy];
    }
    const size_t lds = std::max<size_t>((size_t)std::min(nmax, lds_keys) * sizeof(unsigned long long), getenv("HSCMP_EPI_LDS_KEYS") ? 8192 : 65536);
    A.lds_keys = std::min(nmax, lds_keys); A.lds_bytes = (int)lds;
    const size_t xoff = (size_t)first * T * Fd * esize(level0->dtype);
    if (level0->dtype == HSCMP_F32) {
        auto kern = hier_epilogue_kernel<float>;
        HIP_TRY(last, set_dyn_lds((const void*)kern, lds));
        hipLaunchKernelGGL(kern, dim3(count), dim3(kEpiThreads), lds, last->stream, A, (const float*)((const char*)level0->last_x_dev + xoff));
    } else {
        auto kern = hier_epilogue_kernel<double>;
        HIP_TRY(last, set_dyn_lds((const void*)kern, lds));
        hipLaunchKernelGGL(kern, dim3(count), dim3(kEpiThreads), lds, last->stream, A, (const double*)((const char*)level0->last_x_dev + xoff));
    }
    HIP_TRY(last, hipGetLastError());
    HIP_TRY(last, hipMemcpyAsync(out_n, A.out_n, sizes[2], hipMemcpyDeviceToHost, last->stream));
    HIP_TRY(last, hipMemcpyAsync(out_colptr, A.out_colptr, sizes[3], hipMemcpyDeviceToHost, last->stream));
    if (total > 0) {
        HIP_TRY(last, hipMemcpyAsync(out_indices, A.out_indices, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, last->stream));
        HIP_TRY(last, hipMemcpyAsync(out_data, A.out_data, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, last->stream));
        if (out_events) HIP_TRY(last, hipMemcpyAsync(out_events, A.out_events, (size_t)total * 16, hipMemcpyDeviceToHost, last->stream));
    }
    if (out_residual) HIP_TRY(last, hipMemcpyAsync(out_residual, A.out_residual, nres, hipMemcpyDeviceToHost, last->stream));
    if (out_residual_energy) HIP_TRY(last, hipMemcpyAsync(out_residual_energy, A.out_energy, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, last->stream));
    HIP_TRY(last, hipStreamSynchronize(last->stream));
    return HSCMP_OK;
}

#ifdef HSCMP_DBG_STAMPS
extern "C" int hscmp_debug_blocks(unsigned long long* out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hscmp::g_blk), (size_t)n * 3 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
// diagnostic build only
extern "C" int hscmp_debug_counters(unsigned long long* out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(hscmp::g_cnt), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hscmp::g_cnt), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}

extern "C" int hscmp_debug_stamps(unsigned long long* out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(hscmp::g_stamps), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;      // (64 entries)
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hscmp::g_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

// ---- row-level entry points (modeling.py:899-982 and :1018-1051) -------------------------------------
// One selection (modeling.py:899-982) on a table that is ON THE DEVICE: d_ip [T][K], d_w [K] or null.  Uses the batch
// workspace of the context (a batch held by the context is gone afterwards) but none of its dictionary state.
template <typename R>
static int select_on_device(hscmp_ctx* ctx, const R* d_ip, const R* d_w, int T, int K, int W, int nb_blocks, int offset, double thres,
                            int32_t* out_t, int32_t* out_k, void* out_c, int max_out, int32_t* n_out, const char* who)
{
    hscmp_params hp{};
    hp.nb_nonzero_coefs = -1; hp.nb_blocks = nb_blocks; hp.tolerance_snr = NAN; hp.tolerance_residual_scale = NAN;
    hp.null_coeff_thres = thres; hp.eps = 0.0; hp.max_events = 1; hp.max_rounds = 1;
    DevParams P;
    int rc = make_params_g(ctx, K, W, 1, 1, T, &hp, &P);               // only T, K, W matter for the selection
    if (rc == HSCMP_OK) rc = ensure_workspace_g(ctx, P, false, sizeof(R), false, false);
    if (rc != HSCMP_OK) return rc;
    ctx->have_batch = false;
    ctx->listed_rows = 0;
    State<R> S = make_state<R>(ctx);
    S.D = nullptr; S.Dc = nullptr; S.weights = d_w;
    hipLaunchKernelGGL((table_to_best_kernel<R>), dim3((T + kThreads - 1) / kThreads), dim3(kThreads), 0, ctx->stream,
                       d_ip, T, K, d_w, S.best_c, S.best_k);
    int st[ST_COUNT] = {0};
    st[ST_OFFSET] = offset ? 1 : 0;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stats, st, sizeof(st), hipMemcpyHostToDevice, ctx->stream));
    P.select_only = 1; P.has_snr = 0; P.has_scale = 0;
    set_segments(P, GenericRecorr<R>::kMaxSegments);
    const size_t lds = ((sizeof(typename GenericRecorr<R>::Shared) + 15) / 16) * 16 + GenericRecorr<R>::extra_lds_bytes(P);
    auto kern = iterate_kernel<R, GenericRecorr<R>>;
    HIP_TRY(ctx, set_dyn_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kThreads), lds, ctx->stream, P, S, typename GenericRecorr<R>::Args{});
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(st, ctx->d_stats, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int n = st[ST_EVENTS];
    *n_out = n;
    if (n > max_out) return fail(ctx, HSCMP_ERR_INVALID, "%s: %d atoms selected, room for %d", who, n, max_out);
    if (n > 0) {                                                        // the ordered list lives in the second half of the selection scratch
        HIP_TRY(ctx, hipMemcpyAsync(out_t, ctx->d_sel_t + P.maxsel, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(out_k, ctx->d_sel_k + P.maxsel, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(out_c, (R*)ctx->d_sel_c + P.maxsel, (size_t)n * sizeof(R), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return HSCMP_OK;
}

template <typename R> static int upload_weights(hscmp_ctx* ctx, const void* weights, int K, const R** d_w)
{
    *d_w = nullptr;
    if (!weights) return HSCMP_OK;
    int rc = epi_buffer(ctx, kArenaTabW, (size_t)K * sizeof(R));
    if (rc != HSCMP_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_epi[kArenaTabW], weights, (size_t)K * sizeof(R), hipMemcpyHostToDevice, ctx->stream));
    *d_w = (const R*)ctx->d_epi[kArenaTabW];
    return HSCMP_OK;
}

template <typename R>
static int run_select(hscmp_ctx* ctx, const void* ip, int T, int K, int W, int nb_blocks, int offset, double thres,
                      const void* weights, int32_t* out_t, int32_t* out_k, void* out_c, int max_out, int32_t* n_out)
{
    int rc = epi_buffer(ctx, kArenaRowA, (size_t)T * K * sizeof(R));
    if (rc != HSCMP_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_epi[kArenaRowA], ip, (size_t)T * K * sizeof(R), hipMemcpyHostToDevice, ctx->stream));
    const R* d_w;
    if ((rc = upload_weights<R>(ctx, weights, K, &d_w)) != HSCMP_OK) return rc;
    return select_on_device<R>(ctx, (const R*)ctx->d_epi[kArenaRowA], d_w, T, K, W, nb_blocks, offset, thres, out_t, out_k, out_c, max_out, n_out,
                               "hscmp_select_best_atoms");
}

extern "C" int hscmp_select_best_atoms(hscmp_ctx* ctx, const void* ip, int T, int K, int W, hscmp_dtype dtype, int nb_blocks,
                                       int offset, double null_coeff_thres, const void* weights,
                                       int32_t* out_t, int32_t* out_k, void* out_c, int max_out, int32_t* n_out)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_select_best_atoms: ctx is NULL");
    if (!ip || !out_t || !out_k || !out_c || !n_out || T <= 0 || K <= 0 || W <= 0)
        return fail(ctx, HSCMP_ERR_INVALID, "hscmp_select_best_atoms: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return dtype == HSCMP_F32
        ? run_select<float>(ctx, ip, T, K, W, nb_blocks, offset, null_coeff_thres, weights, out_t, out_k, out_c, max_out, n_out)
        : run_select<double>(ctx, ip, T, K, W, nb_blocks, offset, null_coeff_thres, weights, out_t, out_k, out_c, max_out, n_out);
}

template <typename R> static void launch_update_rows(hscmp_ctx* ctx, const R* d_r, int T, int p, R* d_rows, R* d_table)
{
    const int K = ctx->K, W = ctx->W, nrows = 2 * W - 1;
    DevParams P{};
    P.B = 1; P.T = T; P.K = K; P.W = W; P.F = ctx->F; P.off = (W - 1) / 2;
    const int grid = (nrows * K + kThreads - 1) / kThreads;
    hipLaunchKernelGGL((update_rows_kernel<R>), dim3(grid), dim3(kThreads), 0, ctx->stream, P, d_r, (const R*)ctx->d_D, p, d_rows, d_table);
}

template <typename R> static int run_update_rows(hscmp_ctx* ctx, void* ip, const void* residual, int T, int p)
{
    const int K = ctx->K, W = ctx->W, F = ctx->F, nrows = 2 * W - 1;
    int rc;
    if ((rc = epi_buffer(ctx, kArenaRowA, (size_t)T * F * sizeof(R))) != HSCMP_OK) return rc;
    if ((rc = epi_buffer(ctx, kArenaRowB, (size_t)nrows * K * sizeof(R))) != HSCMP_OK) return rc;
    R* d_r = (R*)ctx->d_epi[kArenaRowA]; R* d_out = (R*)ctx->d_epi[kArenaRowB];
    std::vector<R> rows((size_t)nrows * K);
    HIP_TRY(ctx, hipMemcpyAsync(d_r, residual, (size_t)T * F * sizeof(R), hipMemcpyHostToDevice, ctx->stream));
    launch_update_rows<R>(ctx, d_r, T, p, d_out, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(rows.data(), d_out, rows.size() * sizeof(R), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    R* tab = (R*)ip;
    for (int row = 0; row < nrows; ++row) {               // overlapReplace clipping (utils.py:133-161)
        const int t = p - (W - 1) + row;
        if (t >= 0 && t < T) memcpy(tab + (size_t)t * K, rows.data() + (size_t)row * K, (size_t)K * sizeof(R));
    }
    return HSCMP_OK;
}

extern "C" int hscmp_update_inner_products(hscmp_ctx* ctx, void* ip, const void* residual, int T, int p)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_update_inner_products: ctx is NULL");
    if (ctx->dtype < 0) return fail(ctx, HSCMP_ERR_STATE, "hscmp_update_inner_products: no dictionary set");
    if (!ip || !residual || T <= 0) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_update_inner_products: bad arguments");
    if (p < 0 || p >= T) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_update_inner_products: atom centre %d outside the signal [0, %d)", p, T);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ctx->dtype == HSCMP_F32 ? run_update_rows<float>(ctx, ip, residual, T, p) : run_update_rows<double>(ctx, ip, residual, T, p);
}

// ---- LoCOMP's table on the device (hsc/modeling.py:1267-1425) -----------------------------------------------------------
// The reference keeps innerProducts[T][K] and the residual as host arrays and edits both around every selected atom.
// Here both live in the context's arena: open = :1293 (innerProducts = convolve1d(residual, D, 'same')), select =
// _selectBestAtoms (:899-982) on the resident table, update = the caller's new residual samples + _updateInnerProducts
// (:1018-1051) for every atom of the re-fitted group, in place.  Per iteration the host moves O(W) samples, not T*K.
// Multi-feature tables (hierarchical levels >= 1) are built row by row from the non-zero cells of each row's window
// (table_rows_sparse_kernel); single-feature inputs are dense and keep the dense kernels.  HSCMP_FORCE_DENSE: dense always.
static bool table_rows_are_sparse(const hscmp_ctx* ctx) { return ctx->F > 1 && ctx->W <= 32767 && ctx->F <= 65535 && !getenv("HSCMP_FORCE_DENSE"); }
constexpr int kTableRowCap = 1024;             // listed non-zeros per row window (12 KB of LDS in float64); more: dense chain
template <typename R> static void launch_table_rows_sparse(hscmp_ctx* ctx, const R* d_r, int T, int row0, int nrows, int p, R* d_table)
{
    DevParams P{};
    P.B = 1; P.T = T; P.K = ctx->K; P.W = ctx->W; P.F = ctx->F; P.off = (ctx->W - 1) / 2;
    hipLaunchKernelGGL((table_rows_sparse_kernel<R>), dim3(nrows), dim3(kThreads), (size_t)kTableRowCap * (sizeof(R) + 4), ctx->stream,
                       P, d_r, (const R*)ctx->d_D, row0, p, d_table, kTableRowCap);
}

template <typename R> static int run_table_open(hscmp_ctx* ctx, const void* x, int T)
{
    const int K = ctx->K, F = ctx->F;
    int rc;
    if ((rc = epi_buffer(ctx, kArenaTabRes, (size_t)T * F * sizeof(R))) != HSCMP_OK) return rc;
    if ((rc = epi_buffer(ctx, kArenaTable, (size_t)T * K * sizeof(R))) != HSCMP_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_epi[kArenaTabRes], x, (size_t)T * F * sizeof(R), hipMemcpyHostToDevice, ctx->stream));
    // the caller's buffer may be reused as soon as this returns (as hscmp_table_update promises): wait for the upload (not through
    // ev[]: those are the batch-timing events hscmp_last_kernel_ms reads)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (table_rows_are_sparse(ctx)) launch_table_rows_sparse<R>(ctx, (const R*)ctx->d_epi[kArenaTabRes], T, 0, T, -1, (R*)ctx->d_epi[kArenaTable]);
    else launch_convolve<R>(ctx, (const R*)ctx->d_epi[kArenaTabRes], T, 1, T, (R*)ctx->d_epi[kArenaTable]);
    HIP_TRY(ctx, hipGetLastError());
    ctx->tab_T = T;
    return HSCMP_OK;
}

extern "C" int hscmp_table_open(hscmp_ctx* ctx, const void* x, int T)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_table_open: ctx is NULL");
    if (ctx->dtype < 0) return fail(ctx, HSCMP_ERR_STATE, "hscmp_table_open: no dictionary set");
    if (!x || T <= 0) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_table_open: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->tab_T = 0;
    return ctx->dtype == HSCMP_F32 ? run_table_open<float>(ctx, x, T) : run_table_open<double>(ctx, x, T);
}

extern "C" int hscmp_table_select(hscmp_ctx* ctx, int nb_blocks, int offset, double null_coeff_thres, const void* weights,
                                  int32_t* out_t, int32_t* out_k, void* out_c, int max_out, int32_t* n_out)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_table_select: ctx is NULL");
    if (ctx->tab_T <= 0) return fail(ctx, HSCMP_ERR_STATE, "hscmp_table_select: no table open (hscmp_table_open)");
    if (!out_t || !out_k || !out_c || !n_out) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_table_select: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if (ctx->dtype == HSCMP_F32) {
        const float* d_w;
        if ((rc = upload_weights<float>(ctx, weights, ctx->K, &d_w)) != HSCMP_OK) return rc;
        return select_on_device<float>(ctx, (const float*)ctx->d_epi[kArenaTable], d_w, ctx->tab_T, ctx->K, ctx->W, nb_blocks, offset, null_coeff_thres,
                                       out_t, out_k, out_c, max_out, n_out, "hscmp_table_select");
    }
    const double* d_w;
    if ((rc = upload_weights<double>(ctx, weights, ctx->K, &d_w)) != HSCMP_OK) return rc;
    return select_on_device<double>(ctx, (const double*)ctx->d_epi[kArenaTable], d_w, ctx->tab_T, ctx->K, ctx->W, nb_blocks, offset, null_coeff_thres,
                                    out_t, out_k, out_c, max_out, n_out, "hscmp_table_select");
}

extern "C" int hscmp_table_update(hscmp_ctx* ctx, const void* residual_samples, int start, int count, const int32_t* centres, int ncentres)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_table_update: ctx is NULL");
    if (ctx->tab_T <= 0) return fail(ctx, HSCMP_ERR_STATE, "hscmp_table_update: no table open (hscmp_table_open)");
    const int T = ctx->tab_T;
    if (count < 0 || start < 0 || start + count > T || (count > 0 && !residual_samples) || ncentres < 0 || (ncentres > 0 && !centres))
        return fail(ctx, HSCMP_ERR_INVALID, "hscmp_table_update: bad arguments (start=%d count=%d T=%d)", start, count, T);
    for (int i = 0; i < ncentres; ++i)
        if (centres[i] < 0 || centres[i] >= T) return fail(ctx, HSCMP_ERR_INVALID, "hscmp_table_update: atom centre %d outside the signal [0, %d)", centres[i], T);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t es = esize(ctx->dtype), row = (size_t)ctx->F * es;
    if (count > 0)
        HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->d_epi[kArenaTabRes] + (size_t)start * row, residual_samples, (size_t)count * row, hipMemcpyHostToDevice, ctx->stream));
    const bool sparse_rows_form = table_rows_are_sparse(ctx);
    for (int i = 0; i < ncentres; ++i) {
        if (sparse_rows_form) {
            if (ctx->dtype == HSCMP_F32) launch_table_rows_sparse<float>(ctx, (const float*)ctx->d_epi[kArenaTabRes], T, 0, 2 * ctx->W - 1, centres[i], (float*)ctx->d_epi[kArenaTable]);
            else launch_table_rows_sparse<double>(ctx, (const double*)ctx->d_epi[kArenaTabRes], T, 0, 2 * ctx->W - 1, centres[i], (double*)ctx->d_epi[kArenaTable]);
        } else if (ctx->dtype == HSCMP_F32) launch_update_rows<float>(ctx, (const float*)ctx->d_epi[kArenaTabRes], T, centres[i], nullptr, (float*)ctx->d_epi[kArenaTable]);
        else launch_update_rows<double>(ctx, (const double*)ctx->d_epi[kArenaTabRes], T, centres[i], nullptr, (double*)ctx->d_epi[kArenaTable]);
    }
    HIP_TRY(ctx, hipGetLastError());
    // the host buffer may be reused as soon as this returns
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}

extern "C" int hscmp_table_read(hscmp_ctx* ctx, void* out_table, void* out_residual)
{
    if (!ctx) return fail(nullptr, HSCMP_ERR_INVALID, "hscmp_table_read: ctx is NULL");
    if (ctx->tab_T <= 0) return fail(ctx, HSCMP_ERR_STATE, "hscmp_table_read: no table open (hscmp_table_open)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t es = esize(ctx->dtype);
    if (out_table) HIP_TRY(ctx, hipMemcpyAsync(out_table, ctx->d_epi[kArenaTable], (size_t)ctx->tab_T * ctx->K * es, hipMemcpyDeviceToHost, ctx->stream));
    if (out_residual) HIP_TRY(ctx, hipMemcpyAsync(out_residual, ctx->d_epi[kArenaTabRes], (size_t)ctx->tab_T * ctx->F * es, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HSCMP_OK;
}
