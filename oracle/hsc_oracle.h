/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the convolutional matching-pursuit hot path.
 *
 * Plain-C restatement of the reference algorithm (sbrodeur/hierarchical-sparse-coding,
 * hsc/modeling.py:149-188, 899-1186 and hsc/utils.py:76-161), materialised [T,K] inner-product
 * table and full-table scans included, exactly as the reference does it.  It is the CHECKER for
 * the HIP engine: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  The product (hierarchical-sparse-coding_amd/) never links, imports or calls it.
 *
 * Parity pin: validated against the real reference run in the build container
 * (oracle/ref_loader.py) and against the committed golden vectors in tests/golden/ (generated
 * by tools/make_golden.py from the real reference); see tests/test_oracle_golden.py.
 *
 * Arithmetic that the reference leaves unpinned and the oracle pins (DESIGN.md "Numerics"):
 *   - correlation: c[t,k] = sum_{f,w} xpad[t-off+w, f] * D[k, w, f] as ONE sequential fma chain
 *     starting from +0, f outer / w inner -- the contraction index order f*W + w of the reference's
 *     reshape (modeling.py:181-187).  numpy defers to BLAS, whose order is unspecified; with this
 *     container's OpenBLAS the pinned chain reproduces the reference bit for bit whenever W*F <= ~256;
 *   - energy sums: 256 strided partial sums of rounded squares, then a fixed halving tree
 *     (numpy uses its pairwise summation) -- see hsco_energy_*.
 */
#ifndef HSC_ORACLE_H
#define HSC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* stop reasons, in the order the reference tests them (modeling.py:1125-1158) */
enum {
    HSCO_RUNNING = 0,          /* max_rounds reached, not converged */
    HSCO_STOP_ENERGY_EPS = 1,  /* modeling.py:1125-1130 */
    HSCO_STOP_NNZ = 2,         /* modeling.py:1135-1138 */
    HSCO_STOP_SNR = 3,         /* modeling.py:1139-1142 */
    HSCO_STOP_RESIDUAL_SCALE = 4, /* modeling.py:1145-1148 */
    HSCO_STOP_EMPTY = 5,       /* modeling.py:1150-1153 */
    HSCO_STOP_CALLBACK = 6,    /* modeling.py:1155-1158 (host side) */
    HSCO_STOP_CAPACITY = 7     /* event buffer full (not a reference condition) */
};

typedef struct hsco_params {
    int32_t nb_nonzero_coefs;        /* < 0: None */
    int32_t nb_blocks;               /* 1: single arg-max; > 1: fixed blocks; -1: 'auto' */
    double tolerance_snr;            /* NaN: None */
    double tolerance_residual_scale; /* NaN: None */
    double null_coeff_thres;         /* NaN: None (nothing is dropped) */
    double eps;                      /* finfo(D.dtype).eps, modeling.py:1057 */
    int32_t max_events;              /* capacity of ev_* per signal */
    int32_t max_rounds;              /* <= 0: until converged */
} hsco_params;

/* stats[] layout of hsco_cmp_encode_* */
enum { HSCO_STAT_NNZ = 0, HSCO_STAT_DUPLICATES = 1, HSCO_STAT_ROUNDS = 2, HSCO_STAT_STOP = 3,
       HSCO_STAT_ITERATIONS = 4, HSCO_STAT_COUNT = 8 };

int hsco_version(void);

/* modeling.py:149-188.  x [T,F], D [K,W,F] -> out [Tout,K]; same != 0: zero padded 'same'
 * (Tout = T), else 'valid' (Tout = T-W+1).  Returns Tout or < 0. */
int hsco_convolve1d_f32(const float* x, int T, int F, const float* D, int K, int W, int same, float* out);
int hsco_convolve1d_f64(const double* x, int T, int F, const double* D, int K, int W, int same, double* out);

/* utils.py:76-161: clipped span of a width-W element centred at t: [*start, *end) in the signal
 * and [*estart, *eend) in the element.  Returns the overlap length (0 if none). */
int hsco_span(int T, int W, int t, int* start, int* end, int* estart, int* eend);

/* modeling.py:899-982.  ip [T,K].  Returns the number of atoms written (<= max_out) or < 0. */
int hsco_select_best_atoms_f32(const float* ip, int T, int K, int W, int nb_blocks, int offset,
                               double null_coeff_thres, const float* weights,
                               int32_t* out_t, int32_t* out_k, float* out_c, int max_out);
int hsco_select_best_atoms_f64(const double* ip, int T, int K, int W, int nb_blocks, int offset,
                               double null_coeff_thres, const double* weights,
                               int32_t* out_t, int32_t* out_k, double* out_c, int max_out);

/* modeling.py:1018-1051: re-correlate the 2W-1 rows around centre p (reflect-padded residual
 * span) and replace them in ip [T,K]. */
void hsco_update_inner_products_f32(float* ip, const float* r, int T, int F, const float* D, int K, int W, int p);
void hsco_update_inner_products_f64(double* ip, const double* r, int T, int F, const double* D, int K, int W, int p);

/* energy of v[0..n) in the oracle's pinned order (see header comment) */
float hsco_energy_f32(const float* v, int64_t n);
double hsco_energy_f64(const double* v, int64_t n);

/* modeling.py:1053-1186 (driver) without the CSC epilogue: emits the (t,k,c) events in
 * selection order; the caller accumulates duplicates / clips (oracle/hsc_oracle.py).
 * residual_out [T,F]; energies_out[0] = signal energy, [1] = tracked residual energy.
 * Returns 0 or < 0 on argument / allocation error. */
int hsco_cmp_encode_f32(const float* x, int T, int F, const float* D, int K, int W, const float* weights,
                        const hsco_params* p, int32_t* ev_t, int32_t* ev_k, float* ev_c, int32_t* n_events,
                        float* residual_out, double* energies_out, int32_t* stats);
int hsco_cmp_encode_f64(const double* x, int T, int F, const double* D, int K, int W, const double* weights,
                        const hsco_params* p, int32_t* ev_t, int32_t* ev_k, double* ev_c, int32_t* n_events,
                        double* residual_out, double* energies_out, int32_t* stats);

#ifdef __cplusplus
}
#endif
#endif
