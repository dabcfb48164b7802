/*
 * hscmp.h -- C ABI of the MI355X-native convolutional matching-pursuit engine (libhscmp.so).
 *
 * The reference (sbrodeur/hierarchical-sparse-coding) is pure Python and has no FFI; its plugin
 * seam for this path is the `SparseApproximator` duck type (hsc/modeling.py:657-660):
 * `computeCoefficients(sequence, D, **kw) -> (coefficients, residual)`.  The entry points below
 * are what a ctypes binding of that seam needs (INTEGRATION.md shows the stub); each one names
 * the reference code it replaces.  Plain C types only: pointers, sizes, PODs.  No torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative hscmp_status otherwise; the message is
 *     available from hscmp_last_error(ctx) (thread-local for ctx == NULL failures);
 *   - the caller owns every buffer it passes; the library never keeps a caller pointer after
 *     the call returns (device-pointer entry points: until the work queued on the context's
 *     stream has completed -- hscmp_synchronize);
 *   - a context is bound to one GPU and is not thread-safe; contexts on different GPUs (or
 *     several on one GPU) may be used concurrently from different threads / processes;
 *   - arrays are C-contiguous: signals [B][T][F], dictionary [K][W][F] (reference layout,
 *     modeling.py:1059-1067), events [B][max_events].
 */
#ifndef HSCMP_H
#define HSCMP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSCMP_VERSION 100

typedef struct hscmp_ctx hscmp_ctx;

typedef enum hscmp_status {
    HSCMP_OK = 0,
    HSCMP_ERR_INVALID = -1,      /* bad argument */
    HSCMP_ERR_HIP = -2,          /* HIP runtime error (message has the HIP string) */
    HSCMP_ERR_NO_DEVICE = -3,    /* no gfx950 device / extension cannot run here */
    HSCMP_ERR_STATE = -4,        /* call sequence error (no dictionary, no batch, ...) */
    HSCMP_ERR_UNSUPPORTED = -5,  /* shape outside what the kernels are built for */
    HSCMP_ERR_ALLOC = -6
} hscmp_status;

typedef enum hscmp_dtype { HSCMP_F32 = 0, HSCMP_F64 = 1 } hscmp_dtype;

/* per-signal stop reason, in the order the reference tests them (modeling.py:1125-1158) */
typedef enum hscmp_stop {
    HSCMP_RUNNING = 0,             /* max_rounds reached, not converged: hscmp_continue() resumes */
    HSCMP_STOP_ENERGY_EPS = 1,     /* modeling.py:1125-1130 */
    HSCMP_STOP_NNZ = 2,            /* modeling.py:1135-1138 */
    HSCMP_STOP_SNR = 3,            /* modeling.py:1139-1142 */
    HSCMP_STOP_RESIDUAL_SCALE = 4, /* modeling.py:1145-1148 */
    HSCMP_STOP_EMPTY = 5,          /* modeling.py:1150-1153 */
    HSCMP_STOP_CALLBACK = 6,       /* modeling.py:1155-1158, decided by the host between rounds */
    HSCMP_STOP_CAPACITY = 7,       /* event buffer full: re-run with a larger max_events */
    HSCMP_STOP_STALLED = 8,        /* LoCOMP only, modeling.py:1379-1383: an atom changed the residual energy by less than eps */
    HSCMP_STOP_GROUP = 9           /* LoCOMP only: a neighbourhood beyond the signal's group scratch (more than 511 atoms around one
                                      selection; HSCMP_LOCOMP_GROUP_CAP lowers it), modeling.py:1222-1241; nothing of that atom has been
                                      applied: repeat the signal through the hscmp_table_* loop (or the reference's own LoCOMP) */
} hscmp_stop;

/* which loop hscmp_encode_batch* / hscmp_continue run: ConvolutionalMatchingPursuit.computeCoefficients (modeling.py:1053-1186)
 * or LoCOMP.computeCoefficients (:1267-1425, the joint re-fit of every selected atom with its neighbourhood) */
typedef enum hscmp_method { HSCMP_METHOD_CMP = 0, HSCMP_METHOD_LOCOMP = 1 } hscmp_method;

/* keyword arguments of ConvolutionalMatchingPursuit.computeCoefficients (modeling.py:1053) */
typedef struct hscmp_params {
    int32_t nb_nonzero_coefs;        /* nbNonzeroCoefs; < 0 = None */
    int32_t nb_blocks;               /* nbBlocks: 1, > 1, or -1 = 'auto' (modeling.py:908-918) */
    double tolerance_snr;            /* toleranceSnr in dB; NaN = None */
    double tolerance_residual_scale; /* toleranceResidualScale; NaN = None */
    double null_coeff_thres;         /* minCoefficients as passed to _selectBestAtoms (:1088); NaN = None */
    double eps;                      /* np.finfo(D.dtype).eps (:1057) */
    int32_t max_events;              /* capacity of the per-signal event list */
    int32_t max_rounds;              /* selection rounds per call; <= 0 = until every signal converged */
} hscmp_params;

/* per-signal counters returned by hscmp_fetch_stats: int32 [B][HSCMP_STAT_COUNT] */
enum { HSCMP_STAT_NNZ = 0,        /* distinct non-zero (t,k), modeling.py:1106-1111 */
       HSCMP_STAT_DUPLICATES = 1, /* re-selections of an existing (t,k) */
       HSCMP_STAT_ROUNDS = 2,     /* nbSelections, modeling.py:1160 */
       HSCMP_STAT_STOP = 3,       /* hscmp_stop */
       HSCMP_STAT_ITERATIONS = 4, /* nbIterations = applied atoms, modeling.py:1122 */
       HSCMP_STAT_EVENTS = 5,     /* entries in the event list (== iterations) */
       HSCMP_STAT_SLOTS = 6,      /* distinct (t,k) entries */
       HSCMP_STAT_OFFSET = 7,     /* current half-block offset toggle, modeling.py:1163 */
       HSCMP_STAT_COUNT = 8 };

int hscmp_version(void);

/* One context per GPU (or per host thread).  device_id: HIP ordinal. */
int hscmp_create(hscmp_ctx** out, int device_id);
void hscmp_destroy(hscmp_ctx* ctx);
const char* hscmp_last_error(hscmp_ctx* ctx);

/* Queue all work of this context on `hip_stream` (a hipStream_t; NULL = the context's own
 * stream).  hscmp_synchronize waits for it. */
int hscmp_set_stream(hscmp_ctx* ctx, void* hip_stream);

/* Method of the batch entry points from the next encode on (sticky; default HSCMP_METHOD_CMP).  Replaces the choice of the
 * approximator class, modeling.py:1470-1487 (`method` of HierarchicalConvolutionalMatchingPursuit) / :1191 (class LoCOMP).
 * Under HSCMP_METHOD_LOCOMP the statistic HSCMP_STAT_NNZ counts the stored non-zero coefficients (`coefficients.nnz`,
 * :1368) and coefficients are compared with the reference at the tolerance of its pseudo-inverse, not bit for bit. */
int hscmp_set_method(hscmp_ctx* ctx, int method);
int hscmp_synchronize(hscmp_ctx* ctx);

/* Dictionary D [K][W][F] and optional selection weights [K] (modeling.py:902-906), host
 * pointers; replaces the `D` / `weights` arguments of computeCoefficients (:1053).
 * The dictionary stays resident on the GPU until replaced. */
int hscmp_set_dictionary(hscmp_ctx* ctx, const void* D, int K, int W, int F, hscmp_dtype dtype,
                         const void* weights);

/* modeling.py:149-188 convolve1d(sequence, filters, padding): x [T][F] host -> out [Tout][K]
 * host; same != 0: zero-padded 'same' (Tout = T), else 'valid' (Tout = T-W+1).  Uses the
 * dictionary of the context. */
int hscmp_convolve1d(hscmp_ctx* ctx, const void* x, int T, int same, void* out);

/* ConvolutionalMatchingPursuit._selectBestAtoms (modeling.py:899-982) on a materialised table
 * ip [T][K] (host, dtype): single arg-max (nb_blocks = 1) or blocked selection (nb_blocks > 1,
 * -1 = 'auto'; `offset` = half-block shift) with the null / interference filters and the |c| ordering.
 * weights [K] or NULL; null_coeff_thres NaN = None.  Writes up to max_out atoms (position, atom
 * index, coefficient in dtype) in the reference's output order, *n_out = count. */
int hscmp_select_best_atoms(hscmp_ctx* ctx, const void* ip, int T, int K, int W, hscmp_dtype dtype, int nb_blocks,
                            int offset, double null_coeff_thres, const void* weights,
                            int32_t* out_t, int32_t* out_k, void* out_c, int max_out, int32_t* n_out);

/* ConvolutionalMatchingPursuit._updateInnerProducts (modeling.py:1018-1051) for ONE atom centre p:
 * re-correlates rows p-(W-1) .. p+(W-1) against the reflect-padded residual [T][F] (host) with the
 * context's dictionary and replaces them in ip [T][K] (host, in place). */
int hscmp_update_inner_products(hscmp_ctx* ctx, void* ip, const void* residual, int T, int p);

/* LoCOMP (modeling.py:1267-1425) keeps `innerProducts` [T][K] and the residual as arrays that it edits around every
 * selected atom.  These four entry points keep both ON THE DEVICE, inside the context, so that an iteration moves
 * O(W) samples over PCIe instead of the T*K table:
 *   hscmp_table_open    innerProducts = convolve1d(x, D, padding='same') (:1293); residual := x [T][F] (host, dtype)
 *   hscmp_table_select  _selectBestAtoms (:899-982) on the resident table; arguments as hscmp_select_best_atoms
 *   hscmp_table_update  the residual samples [start, start+count) are replaced by residual_samples [count][F] (host;
 *                       what _updateResidual :996-1016 produced for the re-fitted group), then _updateInnerProducts
 *                       (:1018-1051) runs in place for every atom centre of centres[ncentres]
 *   hscmp_table_read    copies the table [T][K] and / or the residual [T][F] back (either may be NULL): tests, and
 *                       callers that want the reference's arrays
 * The table belongs to the context's dictionary and stays valid until the next hscmp_table_open, hscmp_set_dictionary
 * or hscmp_destroy; selecting on it reuses the batch workspace (a batch held by the context is dropped). */
int hscmp_table_open(hscmp_ctx* ctx, const void* x, int T);
int hscmp_table_select(hscmp_ctx* ctx, int nb_blocks, int offset, double null_coeff_thres, const void* weights,
                       int32_t* out_t, int32_t* out_k, void* out_c, int max_out, int32_t* n_out);
int hscmp_table_update(hscmp_ctx* ctx, const void* residual_samples, int start, int count, const int32_t* centres, int ncentres);
int hscmp_table_read(hscmp_ctx* ctx, void* out_table, void* out_residual);

/* ConvolutionalMatchingPursuit.computeCoefficients (modeling.py:1053-1169) for a batch of B
 * independent signals x [B][T][F] (host memory): initial correlation, then the greedy
 * select / subtract / local re-correlate loop, entirely on the GPU.  Results stay in the
 * context until fetched. */
int hscmp_encode_batch(hscmp_ctx* ctx, const void* x, int B, int T, const hscmp_params* params);

/* Same, x_dev already resident in GPU memory (device pointer, same layout); asynchronous on the
 * context's stream. */
int hscmp_encode_batch_device(hscmp_ctx* ctx, const void* x_dev, int B, int T, const hscmp_params* params);

/* Level chaining of the hierarchical encoder (modeling.py:1489, `input = levelCoefficients.todense()`),
 * entirely on the device: the accumulated coefficients of signals [first, first+count) of `prev`
 * (its distinct (t,k) slots, clipped like the CSC epilogue modeling.py:1171-1181 with
 * min_coefficients; NaN = None) become the dense input [count][T][K_prev] of `ctx`, which is then
 * encoded like hscmp_encode_batch_device.  ctx must hold a float64 dictionary with F == K_prev; both
 * contexts live on the same GPU. */
int hscmp_encode_batch_from_level(hscmp_ctx* ctx, hscmp_ctx* prev, int first, int count, double min_coefficients,
                                  const hscmp_params* params);

/* Window assignment of the convolutional k-means dictionary learner (ConvolutionalDictionaryLearner.
 * _train_kmean, modeling.py:454-460 = convolve1d_batch(windows, D, 'valid') + arg-max of |c| per window):
 * windows [N][L][F] in the dictionary's dtype (host), L >= W.  For every window the flat arg-max over
 * (position 0..L-W, atom) in C order: out_t[n] position, out_k[n] atom, out_c[n] its coefficient (dtype;
 * may be NULL). */
int hscmp_assign_windows(hscmp_ctx* ctx, const void* windows, int N, int L, int32_t* out_t, int32_t* out_k, void* out_c);

/* Host-side synthesis, the sparse branch of reconstructSignal (modeling.py:226-263, used by the residual of the
 * hierarchical encoder, :1596-1611): for every event i in the order given, signal[rows[i] - (W-1)/2 + w][f] +=
 * data[i] * D[cols[i]][w][f], clipped at the borders (utils.py:103-131).  signal float64 [T][Fd] (accumulated in
 * place), D [K][W][Fd] float32 (dict_is_f32 != 0) or float64.  No GPU work and no context: plain host code that
 * can run on many threads at once. */
int hscmp_host_overlap_add(double* signal, int64_t T, int Fd, const int64_t* rows, const int64_t* cols, const double* data,
                           int64_t n, const void* D, int W, int dict_is_f32);

/* Host-side CSC assembly of one signal's coefficient slots (the epilogue of computeCoefficients, modeling.py:
 * 1171-1181): entries that are zero or below min_coefficients (NaN: no clip) are dropped, the rest ordered by
 * (atom, position).  indptr int32 [K+1]; indices int32 / data float64 with room for n entries, of which the first
 * indptr[K] are written.  No GPU work, no context. */
int hscmp_host_slots_to_csc(const int32_t* slot_t, const int32_t* slot_k, const double* slot_a, int64_t n, int K,
                            double min_coefficients, int32_t* indptr, int32_t* indices, double* data);

/* Epilogue of the hierarchical encoder, on the device, for the `count` signals the LAST level's context `last` holds
 * (signals [first, first + count) of the level-0 context `level0`, whose input must still be resident -- it is after
 * hscmp_encode_batch; after hscmp_encode_batch_device the caller's buffer must still be valid).  From the last level's
 * accumulated coefficients (its distinct (t, column) slots) it produces, per signal:
 *   - the CSC form of modeling.py:1171-1181 (zeros and |c| < min_coefficients dropped; NaN: no clip): row indices and
 *     float64 values in column-major order, and column pointers over the last level's columns; level l of the
 *     "distributed" result (convertToDistributedCoefficients, modeling.py:1556-1594) owns columns [col0, col1) of it
 *     under the same column numbers;
 *   - the event records of dataset.py:798-811, (int32 time, int32 level, int32 index, float32 value), sorted by time,
 *     then level, then index (out_events may be NULL);
 *   - the residual of modeling.py:1596-1611, x - sum_l reconstructSignal(level l, levels[l].rep) in float64, every sum
 *     in the order of the reference's sequential overlap-add (bit-identical to hscmp_host_overlap_add level by level;
 *     out_residual [count][T][Fd] may be NULL);
 *   - out_residual_energy [count] (may be NULL): the sum of the squared residual samples of every signal, summed on the
 *     device in a fixed order -- what a caller that only checks the reconstruction quality (10 log10(E_x / E_r)) needs,
 *     without moving T samples per signal over PCIe.
 * levels[l].rep: host pointer to the input-level patterns [>= col1][scale][Fd] of level l (getMultiscaleDictionaries),
 * float32 (rep_is_f32 != 0) or float64.  offsets [count + 1] (host): entry offset of every signal in the packed outputs
 * out_indices / out_data / out_events (offsets[b + 1] - offsets[b] >= that signal's slot count); out_n [count] receives
 * the entries actually written per signal, out_colptr [count][K_last + 1] the column pointers. */
typedef struct hscmp_epilogue_level {
    int32_t col0, col1;    /* columns of the last level's matrix that belong to this level; col1 <= col0: none */
    int32_t scale;         /* taps of the level's input-level patterns */
    int32_t rep_is_f32;
    const void* rep;
} hscmp_epilogue_level;
int hscmp_hierarchy_epilogue(hscmp_ctx* last, hscmp_ctx* level0, int first, const hscmp_epilogue_level* levels, int nlevels,
                             double min_coefficients, const int64_t* offsets, int32_t* out_n, int32_t* out_colptr,
                             int32_t* out_indices, double* out_data, void* out_events, double* out_residual,
                             double* out_residual_energy);

/* Run up to max_rounds further selection rounds on the signals that have not converged
 * (modeling.py:1086 loop); used by hosts that evaluate a stopCondition callback (:1155-1158)
 * between rounds.  max_rounds <= 0: until converged. */
int hscmp_continue(hscmp_ctx* ctx, int max_rounds);

/* Enlarge the per-signal event / slot lists to new_max_events (> the current max_events), keeping
 * their contents, and put the signals that stopped with HSCMP_STOP_CAPACITY back to HSCMP_RUNNING.
 * A round is never started unless all its atoms fit the lists, so hscmp_continue() afterwards
 * reproduces an uninterrupted run bit for bit (the reference's lists are unbounded Python objects,
 * modeling.py:1101-1114). */
int hscmp_grow_events(hscmp_ctx* ctx, int new_max_events);

/* Mark signal b as converged by the host-side stopCondition (modeling.py:1155-1158). */
int hscmp_stop_signal(hscmp_ctx* ctx, int b);

/* Results of the last encode (host buffers, any may be NULL):
 *   ev_t, ev_k  int32 [B][max_events]   selected positions / atom indices IN SELECTION ORDER
 *   ev_c        dtype [B][max_events]   coefficient of each selection (modeling.py:970,946)
 *   stats       int32 [B][HSCMP_STAT_COUNT]
 *   residual    dtype [B][T][F]         (modeling.py:1071,1117)
 *   energies    double [B][2]           signal energy, tracked residual energy (:1070,1014) */
int hscmp_fetch_events(hscmp_ctx* ctx, int32_t* ev_t, int32_t* ev_k, void* ev_c);
int hscmp_fetch_stats(hscmp_ctx* ctx, int32_t* stats);
int hscmp_fetch_residual(hscmp_ctx* ctx, void* residual);
int hscmp_fetch_energies(hscmp_ctx* ctx, double* energies);

/* Accumulated coefficients (the float64 `coefficients[t,k] += c` of modeling.py:1114,:992) as
 * distinct (t,k) slots in first-selection order: slot_t, slot_k int32 [B][max_events],
 * slot_acc double [B][max_events]; counts in stats[HSCMP_STAT_SLOTS]. */
int hscmp_fetch_slots(hscmp_ctx* ctx, int32_t* slot_t, int32_t* slot_k, double* slot_acc);

/* Device pointers of the result arrays (for callers that keep everything on the GPU, e.g. the
 * multi-GPU gather and bench.py); valid until the next encode with a larger shape. */
typedef struct hscmp_device_view {
    int32_t B, T, F, K, W, max_events, dtype, reserved;
    void* ev_t; void* ev_k; void* ev_c; void* stats; void* residual; void* energies;
    void* best_c; void* best_k;   /* per-position best coefficient / atom (table-free state) */
} hscmp_device_view;
int hscmp_get_device_view(hscmp_ctx* ctx, hscmp_device_view* view);

/* Durations (ms, HIP events on the context's stream) of the kernels of the last encode:
 * out[0] = prepare (copy + signal energy), out[1] = initial correlation (modeling.py:1077),
 * out[2] = greedy loop (modeling.py:1086-1163), out[3] = reserved. */
int hscmp_last_kernel_ms(hscmp_ctx* ctx, float* out4);

/* Free / total memory of the context's GPU in bytes (hosts size their signal chunks with it: a level of the
 * hierarchical encoder keeps a dense float64 residual [T][K_prev] per signal on the device). */
int hscmp_mem_info(hscmp_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);

/* Copy `nbytes` of device memory of the context's GPU to the host, ordered behind the context's stream.  For callers that handed
 * the batch over as a device pointer (hscmp_encode_batch_device) and need single signals back on the host -- the per-signal host
 * loop that takes over a signal the batch path gave up on (hsc/modeling.py:1267 with a stopCondition, ...). */
int hscmp_copy_from_device(hscmp_ctx* ctx, const void* src_dev, uint64_t nbytes, void* dst_host);

/* Name of the kernel variant the last encode dispatched ("mfma_f32", "generic_f64", ...). */
const char* hscmp_last_variant(hscmp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
