/*
 * TEST INFRASTRUCTURE ONLY -- type-generic body of the CPU oracle (see hsc_oracle.h).
 * Included twice by hsc_oracle.c with REAL = float / double.
 *
 *   REAL      element type
 *   SFX(n)    n##_f32 / n##_f64
 *   RFMA      fmaf / fma          (the single-rounding multiply-add of the pinned chain)
 *   RABS      fabsf / fabs
 */

/* Pinned correlation order: c = sum_{f,w} win[w,f] * d[w,f] as ONE sequential fma chain from +0,
 * f outer / w inner -- the contraction index order f*W + w of the reference's reshape
 * (modeling.py:181-187). */
/* K independent chains (one per atom) advanced together: acc[k] = fma(win[w,f], Dt[e,k], acc[k])
 * for e = f*W + w ascending.  Per chain this is exactly dot_chain(); the k loop only vectorises. */
static void SFX(row_chains)(const REAL* win, const REAL* Dt, int K, int W, int F, REAL* acc)
{
    for (int k = 0; k < K; ++k) acc[k] = (REAL)0;
    for (int f = 0; f < F; ++f)
        for (int w = 0; w < W; ++w) {
            const REAL xv = win[w * F + f];
            /* a zero sample leaves every chain as it is: fma(+-0, d, acc) == acc for finite d (acc starts at +0 and a
             * round-to-nearest sum never yields -0 from it), so the step is skipped.  The chain ORDER is unchanged;
             * this only makes the almost-all-zero inputs of the hierarchical levels >= 1 (modeling.py:1489) affordable. */
            if (xv == (REAL)0) continue;
            const REAL* d = Dt + (int64_t)(f * W + w) * K;
            for (int k = 0; k < K; ++k) acc[k] = RFMA(xv, d[k], acc[k]);
        }
}

/* Dt[e = f*W + w][k] = D[k][w][f] */
static REAL* SFX(transpose_dict)(const REAL* D, int K, int W, int F)
{
    REAL* Dt = (REAL*)malloc(sizeof(REAL) * (size_t)K * W * F);
    if (!Dt) return NULL;
    for (int k = 0; k < K; ++k)
        for (int w = 0; w < W; ++w)
            for (int f = 0; f < F; ++f)
                Dt[(int64_t)(f * W + w) * K + k] = D[((int64_t)k * W + w) * F + f];
    return Dt;
}

/* modeling.py:149-188 */
int SFX(hsco_convolve1d)(const REAL* x, int T, int F, const REAL* D, int K, int W, int same, REAL* out)
{
    if (T <= 0 || F <= 0 || K <= 0 || W <= 0) return -1;
    int Tout, lead;
    if (same) { Tout = T; lead = (W - 1) / 2; }          /* modeling.py:159-164: (W/2-1, W/2) even, (W/2, W/2) odd */
    else { Tout = T - W + 1; lead = 0; if (Tout <= 0) return -2; }
    /* explicit zero-padded copy, so that padded taps go through the same fma chain */
    const int64_t np_ = (int64_t)(Tout + W - 1) * F;
    REAL* xp = (REAL*)calloc((size_t)np_, sizeof(REAL));
    REAL* Dt = SFX(transpose_dict)(D, K, W, F);
    if (!xp || !Dt) { free(xp); free(Dt); return -3; }
    memcpy(xp + (int64_t)lead * F, x, sizeof(REAL) * (size_t)T * F);
    for (int t = 0; t < Tout; ++t)
        SFX(row_chains)(xp + (int64_t)t * F, Dt, K, W, F, out + (int64_t)t * K);
    free(xp); free(Dt);
    return Tout;
}

/* pinned energy order: 256 strided partials of rounded squares, halving tree per 64, then
 * (P0+P1)+(P2+P3).  Mirrors the HIP block reduction exactly (DESIGN.md "Numerics"). */
REAL SFX(hsco_energy)(const REAL* v, int64_t n)
{
    REAL p[256];
    for (int j = 0; j < 256; ++j) p[j] = (REAL)0;
    for (int64_t i = 0; i < n; ++i) {
        volatile REAL sq = v[i] * v[i];
        p[i & 255] = p[i & 255] + sq;
    }
    for (int w = 0; w < 4; ++w)
        for (int m = 32; m >= 1; m >>= 1)
            for (int j = 0; j < m; ++j) p[64 * w + j] = p[64 * w + j] + p[64 * w + j + m];
    volatile REAL a = p[0] + p[64];
    volatile REAL b = p[128] + p[192];
    return a + b;
}

/* energy of the clipped width-W window centred at t (utils.py:76-101 `peek`, then sum of squares) */
static REAL SFX(window_energy)(const REAL* r, int T, int F, int W, int t)
{
    int s, e, es, ee;
    if (hsco_span(T, W, t, &s, &e, &es, &ee) <= 0) return (REAL)0;
    return SFX(hsco_energy)(r + (int64_t)s * F, (int64_t)(e - s) * F);
}

/* max of |ip[r,k] * w[k]| (or |ip|) over rows [r0, r1); exact whatever the evaluation order */
__attribute__((optimize("O3", "no-trapping-math", "finite-math-only", "no-signed-zeros")))
static REAL SFX(max_abs_score)(const REAL* ip, const REAL* weights, int r0, int r1, int K)
{
    REAL m = (REAL)0;
    if (weights) {
        for (int t = r0; t < r1; ++t) {
            const REAL* row = ip + (int64_t)t * K;
            for (int k = 0; k < K; ++k) { REAL a = RABS(row[k] * weights[k]); m = a > m ? a : m; }
        }
    } else {
        const int64_t n = (int64_t)(r1 - r0) * K;
        const REAL* v = ip + (int64_t)r0 * K;
        for (int64_t i = 0; i < n; ++i) { REAL a = RABS(v[i]); m = a > m ? a : m; }
    }
    return m;
}

/* modeling.py:899-982 */
int SFX(hsco_select_best_atoms)(const REAL* ip, int T, int K, int W, int nb_blocks, int offset,
                                double null_coeff_thres, const REAL* weights,
                                int32_t* out_t, int32_t* out_k, REAL* out_c, int max_out)
{
    const int has_thres = !isnan(null_coeff_thres);
    if (nb_blocks == 1) {
        /* modeling.py:965-975: flat arg-max of |scores| in C order (ties -> smallest t, then k):
         * per-chunk maxima (order-free, vectorised), then the first element of the first chunk
         * that attains the global maximum */
        enum { CH = 64 };
        REAL best = (REAL)0;
        int bchunk = 0;
        for (int r0 = 0, ci = 0; r0 < T; r0 += CH, ++ci) {
            const REAL m = SFX(max_abs_score)(ip, weights, r0, r0 + CH < T ? r0 + CH : T, K);
            if (m > best) { best = m; bchunk = ci; }
        }
        int64_t bi = (int64_t)bchunk * CH * K;
        const int64_t n = (int64_t)T * K;
        if (weights) {
            for (int64_t i = bi; i < n; ++i) {
                volatile REAL s = ip[i] * weights[i % K];
                if (RABS(s) == best) { bi = i; break; }
            }
        } else {
            for (int64_t i = bi; i < n; ++i) if (RABS(ip[i]) == best) { bi = i; break; }
        }
        REAL c = ip[bi];
        if (has_thres && !((double)RABS(c) > null_coeff_thres)) return 0;
        if (max_out < 1) return -4;
        out_t[0] = (int32_t)(bi / K); out_k[0] = (int32_t)(bi % K); out_c[0] = c;
        return 1;
    }

    /* modeling.py:908-963 */
    int bs;
    if (nb_blocks < 0) bs = 4 * W;                       /* 'auto', :912 */
    else bs = (int)floor((double)T / (double)nb_blocks); /* :914 */
    if (bs % 2 == 1) bs += 1;                            /* :916-917 */
    if (bs <= 0) return -5;
    int nb = (int)ceil((double)T / (double)bs);          /* :918 */
    int pad0 = 0;
    if (offset) { pad0 = bs / 2; nb += 1; }              /* :922-925 */

    int32_t* t = (int32_t*)malloc(sizeof(int32_t) * (size_t)nb);
    int32_t* f = (int32_t*)malloc(sizeof(int32_t) * (size_t)nb);
    REAL* c = (REAL*)malloc(sizeof(REAL) * (size_t)nb);
    int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)nb);
    if (!t || !f || !c || !keep) { free(t); free(f); free(c); free(keep); return -3; }

    int n = 0;
    for (int j = 0; j < nb; ++j) {
        /* :935-937: flat arg-max over the [bs,K] window, zero padded rows included */
        REAL best = (REAL)-1; int brel = 0, bk = 0;
        for (int rel = 0; rel < bs; ++rel) {
            const int tt = j * bs + rel - pad0;
            if (tt < 0 || tt >= T) {
                if ((REAL)0 > best) { best = (REAL)0; brel = rel; bk = 0; }
                continue;
            }
            const REAL* row = ip + (int64_t)tt * K;
            for (int k = 0; k < K; ++k) {
                REAL a;
                if (weights) { volatile REAL s = row[k] * weights[k]; a = RABS(s); }
                else a = RABS(row[k]);
                if (a > best) { best = a; brel = rel; bk = k; }
            }
        }
        const int tt = j * bs + brel - pad0;
        if (tt < 0 || tt > T - 1) continue;              /* :940-942 */
        const REAL cc = ip[(int64_t)tt * K + bk];        /* :946 unweighted coefficient */
        if (has_thres && !((double)RABS(cc) > null_coeff_thres)) continue; /* :947-948 */
        t[n] = tt; f[n] = bk; c[n] = cc; ++n;
    }

    /* :951-957 interference filter: vs the UNFILTERED predecessor, first always kept, and the
     * whole filter is skipped when no gap qualifies */
    if (n > 1) {
        int nk = 0;
        for (int i = 1; i < n; ++i) if (t[i] - t[i - 1] >= W) keep[nk++] = i;
        if (nk > 0) {
            int m = 1; /* index 0 kept */
            for (int q = 0; q < nk; ++q) { int i = keep[q]; t[m] = t[i]; f[m] = f[i]; c[m] = c[i]; ++m; }
            n = m;
        }
    }

    /* :960-962 argsort(|c|)[::-1]: descending; among equal |c| the later entry first */
    for (int i = 0; i < n; ++i) keep[i] = i;
    for (int i = 1; i < n; ++i) {          /* stable ascending insertion sort of indices by |c| */
        int idx = keep[i]; REAL a = RABS(c[idx]); int j = i - 1;
        while (j >= 0 && RABS(c[keep[j]]) > a) { keep[j + 1] = keep[j]; --j; }
        keep[j + 1] = idx;
    }
    int ret = n;
    if (n > max_out) ret = -4;
    else for (int i = 0; i < n; ++i) { int idx = keep[n - 1 - i]; out_t[i] = t[idx]; out_k[i] = f[idx]; out_c[i] = c[idx]; }
    free(t); free(f); free(c); free(keep);
    return ret;
}

/* modeling.py:1018-1051 with the dictionary already transposed (Dt[e][k]) */
static void SFX(update_inner_products_t)(REAL* ip, const REAL* r, int T, int F, const REAL* Dt, int K, int W, int p)
{
    const int off = (W - 1) / 2;
    const int tstart = p - off - (W - 1);               /* :1028-1033 */
    const int tend = p + W / 2 + (W - 1);               /* :1038 */
    const int sidx = tstart < 0 ? 0 : tstart;           /* :1034 */
    const int eidx = tend > T - 1 ? T - 1 : tend;       /* :1039 */
    const int n = eidx - sidx + 1;                      /* slice that np.pad(mode='reflect') extends, :1046 */
    const int span = 3 * W - 2;
    REAL* pad = (REAL*)malloc(sizeof(REAL) * (size_t)span * F);
    for (int j = 0; j < span; ++j) {
        int li = tstart + j - sidx;                     /* index relative to the slice */
        int m;
        if (n == 1) m = 0;
        else {
            const int period = 2 * (n - 1);
            m = li % period; if (m < 0) m += period;
            if (m >= n) m = period - m;
        }
        memcpy(pad + (int64_t)j * F, r + (int64_t)(sidx + m) * F, sizeof(REAL) * F);
    }
    for (int j = 0; j < 2 * W - 1; ++j) {               /* 'valid' rows, :1047-1049 */
        const int t = p - (W - 1) + j;
        if (t < 0 || t >= T) continue;                  /* overlapReplace clipping, utils.py:133-161 */
        SFX(row_chains)(pad + (int64_t)j * F, Dt, K, W, F, ip + (int64_t)t * K);
    }
    free(pad);
}

/* modeling.py:1018-1051 */
void SFX(hsco_update_inner_products)(REAL* ip, const REAL* r, int T, int F, const REAL* D, int K, int W, int p)
{
    REAL* Dt = SFX(transpose_dict)(D, K, W, F);
    if (!Dt) return;
    SFX(update_inner_products_t)(ip, r, T, F, Dt, K, W, p);
    free(Dt);
}

/* modeling.py:1053-1169 (the CSC epilogue :1171-1181 is host-side, oracle/hsc_oracle.py) */
int SFX(hsco_cmp_encode)(const REAL* x, int T, int F, const REAL* D, int K, int W, const REAL* weights,
                         const hsco_params* p, int32_t* ev_t, int32_t* ev_k, REAL* ev_c, int32_t* n_events,
                         REAL* residual_out, double* energies_out, int32_t* stats)
{
    if (T <= 0 || F <= 0 || K <= 0 || W <= 0 || !p) return -1;
    const int WF = W * F;
    const int64_t TF = (int64_t)T * F;
    REAL* r = residual_out;
    memcpy(r, x, sizeof(REAL) * (size_t)TF);                          /* :1071 */
    REAL* ip = (REAL*)malloc(sizeof(REAL) * (size_t)T * K);
    REAL* Dt = SFX(transpose_dict)(D, K, W, F);
    /* worst case atoms per round: number of blocks + 1 */
    int max_sel = 1;
    if (p->nb_blocks != 1) {
        int bs = p->nb_blocks < 0 ? 4 * W : (int)floor((double)T / (double)p->nb_blocks);
        if (bs % 2 == 1) bs += 1;
        if (bs <= 0) { free(ip); free(Dt); return -5; }
        max_sel = (int)ceil((double)T / (double)bs) + 1;
    }
    int32_t* sel_t = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_sel);
    int32_t* sel_k = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_sel);
    REAL* sel_c = (REAL*)malloc(sizeof(REAL) * (size_t)max_sel);
    /* distinct (t,k) slots with their float64 accumulated coefficient (lil_matrix, :1074) */
    const int cap = p->max_events > 0 ? p->max_events : 1;
    int32_t* slot_t = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t* slot_k = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    double* slot_a = (double*)malloc(sizeof(double) * (size_t)cap);
    if (!ip || !Dt || !sel_t || !sel_k || !sel_c || !slot_t || !slot_k || !slot_a) {
        free(ip); free(Dt); free(sel_t); free(sel_k); free(sel_c); free(slot_t); free(slot_k); free(slot_a);
        return -3;
    }
    int nslots = 0;

    const REAL energy_signal = SFX(hsco_energy)(x, TF);               /* :1070 */
    REAL energy_residual = energy_signal;                             /* :1072 */
    SFX(hsco_convolve1d)(r, T, F, D, K, W, 1, ip);                    /* :1077 */

    const int has_snr = !isnan(p->tolerance_snr);
    const int has_scale = !isnan(p->tolerance_residual_scale);
    /* snr >= tol  <=>  Esig/Eres >= 10^(tol/10)  (:1132,1139 without the log) */
    const double snr_ratio = has_snr ? pow(10.0, p->tolerance_snr / 10.0) : 0.0;

    int nnz = 0, ndup = 0, niter = 0, nrounds = 0, nev = 0;
    int offset = 0, converged = 0, stop = HSCO_RUNNING;

    while (!converged) {
        if (p->max_rounds > 0 && nrounds >= p->max_rounds) break;
        int nsel = SFX(hsco_select_best_atoms)(ip, T, K, W, p->nb_blocks, offset, p->null_coeff_thres,
                                               weights, sel_t, sel_k, sel_c, max_sel);   /* :1088 */
        if (nsel < 0) { stop = -100 + nsel; break; }

        if (has_snr && nsel > 1) {                                    /* weak-atom filter :1090-1099 */
            /* energySignal / 10^(snr/10) in REAL (python float is weak under NEP 50), then / prod(shape) in double */
            const REAL tol_energy = energy_signal / (REAL)snr_ratio;
            const double thr = (double)tol_energy / (double)TF;
            int m = 0;
            for (int i = 0; i < nsel; ++i) {
                int s, e, es, ee;
                const int len = hsco_span(T, W, sel_t[i], &s, &e, &es, &ee);
                const REAL mean = SFX(window_energy)(r, T, F, W, sel_t[i]) / (REAL)((int64_t)len * F);
                if ((double)mean >= thr) { sel_t[m] = sel_t[i]; sel_k[m] = sel_k[i]; sel_c[m] = sel_c[i]; ++m; }
            }
            nsel = m;
        }

        for (int i = 0; i < nsel; ++i) {                              /* :1101 */
            if (nev >= p->max_events) { converged = 1; stop = HSCO_STOP_CAPACITY; break; }
            const int t = sel_t[i], k = sel_k[i];
            const REAL c = sel_c[i];
            /* :1106-1111 duplicate / nnz bookkeeping on the accumulated coefficient */
            int si = -1;
            for (int q = 0; q < nslots; ++q) if (slot_t[q] == t && slot_k[q] == k) { si = q; break; }
            if (si >= 0 && fabs(slot_a[si]) > 0.0) ndup += 1;
            else if (RABS(c) > (REAL)0) nnz += 1;
            if (si < 0) { si = nslots++; slot_t[si] = t; slot_k[si] = k; slot_a[si] = 0.0; }
            slot_a[si] += (double)c;                                  /* :1114, :992 */
            ev_t[nev] = t; ev_k[nev] = k; ev_c[nev] = c; ++nev;

            /* :1117, :996-1016 residual update with local energy bookkeeping */
            const REAL e_before = SFX(window_energy)(r, T, F, W, t);
            {
                int s, e, es, ee;
                if (hsco_span(T, W, t, &s, &e, &es, &ee) > 0) {
                    const REAL nc = -c;
                    const REAL* dk = D + (int64_t)k * WF;
                    for (int q = 0; q < (e - s) * F; ++q) {
                        volatile REAL prod = nc * dk[(int64_t)es * F + q];   /* -c*D[k] rounded, then += (utils.py:120,129) */
                        r[(int64_t)s * F + q] = r[(int64_t)s * F + q] + prod;
                    }
                }
            }
            const REAL e_after = SFX(window_energy)(r, T, F, W, t);
            {
                volatile REAL loss = e_before - e_after;              /* :1005 */
                energy_residual = energy_residual - loss;             /* :1014 */
            }
            SFX(update_inner_products_t)(ip, r, T, F, Dt, K, W, t);   /* :1120 */
            niter += 1;

            if ((double)energy_residual < p->eps) { converged = 1; stop = HSCO_STOP_ENERGY_EPS; break; }   /* :1125 */
            if (p->nb_nonzero_coefs >= 0 && nnz >= p->nb_nonzero_coefs) { converged = 1; stop = HSCO_STOP_NNZ; break; } /* :1135 */
            if (has_snr) {
                const REAL q = energy_signal / energy_residual;
                if ((double)q >= snr_ratio) { converged = 1; stop = HSCO_STOP_SNR; break; }              /* :1139 */
            }
        }

        if (has_scale) {                                              /* :1145-1148 */
            REAL mx = (REAL)0;
            for (int64_t q = 0; q < TF; ++q) { REAL a = RABS(r[q]); if (a > mx) mx = a; }
            if ((double)mx <= p->tolerance_residual_scale) { converged = 1; if (stop == HSCO_RUNNING) stop = HSCO_STOP_RESIDUAL_SCALE; }
        }
        if (nsel == 0) { converged = 1; if (stop == HSCO_RUNNING) stop = HSCO_STOP_EMPTY; }             /* :1150-1153 */
        nrounds += 1;                                                 /* :1160 */
        offset = !offset;                                             /* :1163 */
    }

    *n_events = nev;
    energies_out[0] = (double)energy_signal;
    energies_out[1] = (double)energy_residual;
    for (int i = 0; i < HSCO_STAT_COUNT; ++i) stats[i] = 0;
    stats[HSCO_STAT_NNZ] = nnz; stats[HSCO_STAT_DUPLICATES] = ndup; stats[HSCO_STAT_ROUNDS] = nrounds;
    stats[HSCO_STAT_STOP] = stop; stats[HSCO_STAT_ITERATIONS] = niter;
    free(ip); free(Dt); free(sel_t); free(sel_k); free(sel_c); free(slot_t); free(slot_k); free(slot_a);
    return 0;
}
