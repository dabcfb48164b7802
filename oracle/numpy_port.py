"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's CPU path, for timing.

This is the `cpu_baseline` ("kind": "port") of bench.py: the same algorithm AND the same
implementation strategy as the reference's ConvolutionalMatchingPursuit (hsc/modeling.py:1053-1186)
-- strided-window GEMM for the correlation (:181-187), a materialised [T,K] table, a full
`np.abs` + `np.argmax` scan per selection (:967), reflect-padded local re-correlation (:1018-1051),
scipy lil_matrix bookkeeping (:1074,:1106-1114) -- written for Python 3 / current NumPy.  It is what
"the repo's own NumPy CPU path" costs on a given host.  Only single arg-max selection
(nbBlocks=1), the mode BASELINE.json's headline config uses, is restated here; the C oracle
(hsc_oracle.c) covers every mode.

Validated against the golden vectors of the real reference in tests/test_numpy_port.py.
Never imported by the product package.
"""
import numpy as np
import scipy.sparse
from numpy.lib.stride_tricks import as_strided


def _span(T, W, t):
    lo = t - (W - 1) // 2
    hi = t + W // 2 + 1
    s, e = max(0, lo), min(T, hi)
    return s, e, s - lo, W - (hi - e)


def correlate(sequence, filters, padding='valid'):
    """modeling.py:149-188: [T,F] x [K,W,F] -> [Tout,K] through a strided [Tout, F, W] view and one GEMM."""
    W = filters.shape[1]
    if padding == 'same':
        sequence = np.pad(sequence, [((W - 1) // 2, W // 2), (0, 0)], mode='constant')
    windows = as_strided(sequence, shape=(sequence.shape[0] - W + 1, sequence.shape[1], W),
                         strides=(sequence.strides[0], sequence.strides[1], sequence.strides[0]))
    n = int(np.prod(filters.shape[1:]))
    return np.dot(windows.reshape((windows.shape[0], n)), filters.T.reshape(n, filters.shape[0]))


def cmp_encode(sequence, D, nbNonzeroCoefs=None, toleranceSnr=None, toleranceResidualScale=None, minCoefficients=1e-16):
    """modeling.py:1053-1186 with nbBlocks=1.  Returns (csc float64, residual, ordered (t,k,c) trace)."""
    squeeze = sequence.ndim == 1 or D.ndim == 2
    if sequence.ndim == 1:
        sequence = sequence[:, np.newaxis]
    if D.ndim == 2:
        D = D[:, :, np.newaxis]
    eps = np.finfo(D.dtype).eps
    T, K, W = sequence.shape[0], D.shape[0], D.shape[1]
    energy_signal = np.sum(np.square(sequence))
    residual = np.copy(sequence)
    energy_residual = energy_signal
    coefficients = scipy.sparse.lil_matrix((T, K))
    table = correlate(residual, D, padding='same')                       # :1077
    trace_t, trace_k, trace_c = [], [], []
    nnz = 0
    while True:
        t, k = np.unravel_index(np.argmax(np.abs(table)), table.shape)   # :967 full scan, every selection
        c = table[t, k]
        if not (np.abs(c) > minCoefficients):                            # :974
            break
        if np.abs(coefficients[t, k]) > 0.0:                             # :1106-1111
            pass
        elif np.abs(c) > 0.0:
            nnz += 1
        coefficients[t, k] += c                                          # :992
        trace_t.append(int(t)); trace_k.append(int(k)); trace_c.append(c)
        s, e, es, ee = _span(T, W, t)                                    # :996-1016
        before = np.sum(np.square(residual[s:e]))
        residual[s:e] += (-c * D[k])[es:ee]
        after = np.sum(np.square(residual[s:e]))
        energy_residual -= (before - after)
        # :1018-1051 local re-correlation on the reflect-padded span
        tstart = t - (W - 1) // 2 - (W - 1)
        tend = t + W // 2 + (W - 1)
        s0, e0 = max(0, tstart), min(T - 1, tend)
        padded = np.pad(residual[s0:e0 + 1], [(s0 - tstart, tend - e0), (0, 0)], mode='reflect')
        local = correlate(padded, D, padding='valid')
        ls, le, les, lee = _span(T, 2 * W - 1, t)
        table[ls:le] = local[les:lee]                                    # overlapReplace :1049
        if energy_residual < eps:
            break
        if nbNonzeroCoefs is not None and nnz >= nbNonzeroCoefs:
            break
        if toleranceSnr is not None and 10.0 * np.log10(energy_signal / energy_residual) >= toleranceSnr:
            break
        if toleranceResidualScale is not None and np.max(np.abs(residual)) <= toleranceResidualScale:
            break
    if minCoefficients is not None:                                      # :1171-1177
        cx = coefficients.tocoo()
        keep = np.abs(cx.data) >= minCoefficients
        clipped = scipy.sparse.lil_matrix((T, K))
        clipped[cx.row[keep], cx.col[keep]] = cx.data[keep]
        coefficients = clipped
    coefficients = coefficients.tocsc()
    coefficients.eliminate_zeros()
    if squeeze:
        residual = np.squeeze(residual, axis=1)
    return coefficients, residual, (np.array(trace_t, dtype=np.int32), np.array(trace_k, dtype=np.int32), np.array(trace_c))


def _timed_worker(args):
    """Encode signals [first, first+count) of a synthetic batch; returns (selections, seconds)."""
    import time
    import os
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    D, signals, L0 = args
    t0 = time.perf_counter()
    n = 0
    for x in signals:
        _, _, trace = cmp_encode(x, D, nbNonzeroCoefs=L0)
        n += len(trace[0])
    return n, time.perf_counter() - t0
