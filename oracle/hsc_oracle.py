"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/hsc_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does.  See oracle/hsc_oracle.h for what is restated and how the
oracle is pinned against the real reference.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.sparse

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libhsc_oracle.so')

STOP_NAMES = {0: 'running', 1: 'energy_eps', 2: 'nnz', 3: 'snr', 4: 'residual_scale', 5: 'empty',
              6: 'callback', 7: 'capacity'}


class HscoParams(ctypes.Structure):
    _fields_ = [('nb_nonzero_coefs', ctypes.c_int32),
                ('nb_blocks', ctypes.c_int32),
                ('tolerance_snr', ctypes.c_double),
                ('tolerance_residual_scale', ctypes.c_double),
                ('null_coeff_thres', ctypes.c_double),
                ('eps', ctypes.c_double),
                ('max_events', ctypes.c_int32),
                ('max_rounds', ctypes.c_int32)]


_lib = None


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    if force or not os.path.isfile(_LIB_PATH):
        subprocess.check_call(['make', '-C', _HERE] + (['-B'] if force else []),
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.hsco_version.restype = ctypes.c_int
        _lib.hsco_energy_f32.restype = ctypes.c_float
        _lib.hsco_energy_f64.restype = ctypes.c_double
        for n in ('hsco_convolve1d', 'hsco_select_best_atoms', 'hsco_cmp_encode'):
            for s in ('_f32', '_f64'):
                getattr(_lib, n + s).restype = ctypes.c_int
        for s in ('_f32', '_f64'):
            getattr(_lib, 'hsco_update_inner_products' + s).restype = None
        _lib.hsco_span.restype = ctypes.c_int
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return '_f32'
    if dtype == np.float64:
        return '_f64'
    raise TypeError('oracle supports float32 / float64 only, got %s' % dtype)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _nb_blocks_code(nbBlocks):
    if nbBlocks == 'auto':
        return -1
    nb = int(nbBlocks)
    if nb < 1:
        raise ValueError('nbBlocks must be >= 1 or "auto"')
    return nb


def convolve1d(sequence, filters, padding='valid'):
    """modeling.py:149-188 in the oracle's pinned summation order."""
    dtype = np.result_type(sequence.dtype, filters.dtype)
    x = np.ascontiguousarray(np.atleast_2d(sequence).reshape((sequence.shape[0], -1)), dtype=dtype)
    D = np.ascontiguousarray(filters, dtype=dtype)
    K, W = D.shape[0], D.shape[1]
    F = 1 if D.ndim == 2 else D.shape[2]
    assert F == x.shape[1]
    T = x.shape[0]
    if padding == 'same':
        same, Tout = 1, T
    elif padding == 'valid':
        same, Tout = 0, T - W + 1
    else:
        raise Exception('Padding not supported: %s' % (padding))
    out = np.empty((Tout, K), dtype=dtype)
    rc = getattr(lib(), 'hsco_convolve1d' + _sfx(dtype))(_ptr(x), T, F, _ptr(D), K, W, same, _ptr(out))
    if rc < 0:
        raise RuntimeError('hsco_convolve1d failed: %d' % rc)
    return out


def span(T, W, t):
    s, e, es, ee = (ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int())
    n = lib().hsco_span(T, W, t, ctypes.byref(s), ctypes.byref(e), ctypes.byref(es), ctypes.byref(ee))
    return n, s.value, e.value, es.value, ee.value


def peek(signal, width, t):
    """utils.py:76-101"""
    n, s, e, _, _ = span(signal.shape[0], width, t)
    return signal[s:e] if n > 0 else np.array([], dtype=signal.dtype)


def overlapAdd(signal, element, t, copy=False):
    """utils.py:103-131"""
    if copy:
        signal = np.copy(signal)
    n, s, e, es, ee = span(signal.shape[0], element.shape[0], t)
    if n > 0:
        signal[s:e] += element[es:ee]
    return signal


def overlapReplace(signal, element, t, copy=False):
    """utils.py:133-161"""
    if copy:
        signal = np.copy(signal)
    n, s, e, es, ee = span(signal.shape[0], element.shape[0], t)
    if n > 0:
        signal[s:e] = element[es:ee]
    return signal


def select_best_atoms(innerProducts, filterWidth, nbBlocks=1, offset=False, nullCoeffThres=0.0, weights=None):
    """modeling.py:899-982; returns (t, k, c) arrays in the reference's output order."""
    ip = np.ascontiguousarray(innerProducts)
    sfx = _sfx(ip.dtype)
    T, K = ip.shape
    w = None if weights is None else np.ascontiguousarray(weights, dtype=ip.dtype)
    cap = T + 2
    t = np.empty(cap, dtype=np.int32)
    k = np.empty(cap, dtype=np.int32)
    c = np.empty(cap, dtype=ip.dtype)
    thres = float('nan') if nullCoeffThres is None else float(nullCoeffThres)
    n = getattr(lib(), 'hsco_select_best_atoms' + sfx)(
        _ptr(ip), T, K, int(filterWidth), _nb_blocks_code(nbBlocks), int(bool(offset)),
        ctypes.c_double(thres), None if w is None else _ptr(w), _ptr(t), _ptr(k), _ptr(c), cap)
    if n < 0:
        raise RuntimeError('hsco_select_best_atoms failed: %d' % n)
    return t[:n].copy(), k[:n].copy(), c[:n].copy()


def update_inner_products(innerProducts, residual, D, position):
    """modeling.py:1018-1051 for one atom centre, in place on innerProducts."""
    sfx = _sfx(innerProducts.dtype)
    r = np.ascontiguousarray(residual.reshape((residual.shape[0], -1)), dtype=innerProducts.dtype)
    Dc = np.ascontiguousarray(D, dtype=innerProducts.dtype)
    T, F = r.shape
    K, W = Dc.shape[0], Dc.shape[1]
    assert innerProducts.flags.c_contiguous and innerProducts.shape == (T, K)
    getattr(lib(), 'hsco_update_inner_products' + sfx)(_ptr(innerProducts), _ptr(r), T, F, _ptr(Dc), K, W, int(position))
    return innerProducts


def energy(v):
    v = np.ascontiguousarray(v).ravel()
    fn = getattr(lib(), 'hsco_energy' + _sfx(v.dtype))
    return v.dtype.type(fn(_ptr(v), ctypes.c_int64(v.size)))


def events_to_csc(ev_t, ev_k, ev_c, shape, minCoefficients=1e-16):
    """modeling.py:1114 (`+=` into a float64 lil_matrix) and the epilogue :1171-1181."""
    acc = {}
    for t, k, c in zip(ev_t.tolist(), ev_k.tolist(), ev_c.tolist()):
        key = (t, k)
        acc[key] = acc.get(key, 0.0) + float(c)
    rows, cols, data = [], [], []
    for (t, k), v in acc.items():
        if minCoefficients is not None and not (abs(v) >= minCoefficients):
            continue
        if v == 0.0:
            continue
        rows.append(t); cols.append(k); data.append(v)
    m = scipy.sparse.csc_matrix((np.array(data, dtype=np.float64),
                                 (np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64))), shape=shape)
    m.sort_indices()
    return m


def cmp_encode(sequence, D, nbNonzeroCoefs=None, toleranceResidualScale=None, toleranceSnr=None, nbBlocks=1,
               minCoefficients=1e-16, weights=None, maxEvents=None, maxRounds=0):
    """modeling.py:1053-1186.  Returns (csc float64 [T,K], residual, info dict with the ordered trace)."""
    assert sequence.ndim == 1 or sequence.ndim == 2
    assert D.ndim == 2 or D.ndim == 3
    eps = float(np.finfo(D.dtype).eps)
    dtype = np.result_type(sequence.dtype, D.dtype)
    sfx = _sfx(dtype)
    squeeze = (sequence.ndim == 1) or (D.ndim == 2)
    x = np.ascontiguousarray(sequence.reshape((sequence.shape[0], -1)), dtype=dtype)
    Dc = np.ascontiguousarray(D.reshape((D.shape[0], D.shape[1], -1)), dtype=dtype)
    T, F = x.shape
    K, W = Dc.shape[0], Dc.shape[1]
    assert Dc.shape[2] == F
    w = None if weights is None else np.ascontiguousarray(weights, dtype=dtype)
    if maxEvents is None:
        maxEvents = 4 * (nbNonzeroCoefs if nbNonzeroCoefs is not None else 0) + 65536
    params = HscoParams(
        nb_nonzero_coefs=-1 if nbNonzeroCoefs is None else int(nbNonzeroCoefs),
        nb_blocks=_nb_blocks_code(nbBlocks),
        tolerance_snr=float('nan') if toleranceSnr is None else float(toleranceSnr),
        tolerance_residual_scale=float('nan') if toleranceResidualScale is None else float(toleranceResidualScale),
        null_coeff_thres=float('nan') if minCoefficients is None else float(minCoefficients),
        eps=eps, max_events=int(maxEvents), max_rounds=int(maxRounds))
    ev_t = np.empty(maxEvents, dtype=np.int32)
    ev_k = np.empty(maxEvents, dtype=np.int32)
    ev_c = np.empty(maxEvents, dtype=dtype)
    nev = ctypes.c_int32(0)
    residual = np.empty((T, F), dtype=dtype)
    energies = np.zeros(2, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int32)
    rc = getattr(lib(), 'hsco_cmp_encode' + sfx)(
        _ptr(x), T, F, _ptr(Dc), K, W, None if w is None else _ptr(w), ctypes.byref(params),
        _ptr(ev_t), _ptr(ev_k), _ptr(ev_c), ctypes.byref(nev), _ptr(residual), _ptr(energies), _ptr(stats))
    if rc != 0:
        raise RuntimeError('hsco_cmp_encode failed: %d' % rc)
    n = nev.value
    ev_t, ev_k, ev_c = ev_t[:n].copy(), ev_k[:n].copy(), ev_c[:n].copy()
    coefficients = events_to_csc(ev_t, ev_k, ev_c, (T, K), minCoefficients)
    if squeeze:
        residual = np.squeeze(residual, axis=1)
    if residual.dtype != sequence.dtype:
        residual = residual.astype(sequence.dtype)
    info = dict(t=ev_t, k=ev_k, c=ev_c, nnz=int(stats[0]), duplicates=int(stats[1]), rounds=int(stats[2]),
                stop=STOP_NAMES.get(int(stats[3]), int(stats[3])), iterations=int(stats[4]),
                energy_signal=float(energies[0]), energy_residual=float(energies[1]))
    return coefficients, residual, info
