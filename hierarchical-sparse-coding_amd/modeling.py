"""Drop-in for the matching-pursuit path of the reference's `hsc.modeling` (same class / method
names, argument meaning, return types and error behaviour), running on MI355X through
libhscmp.so (include/hscmp.h).

Reference surface reproduced here (hsc/modeling.py):
  convolve1d (:149-188), reconstructSignal (:226-263), Atom (:840-864),
  ConvolutionalMatchingPursuit.computeCoefficients (:1053-1186),
  ConvolutionalSparseCoder (:1656-1669).
The hierarchical encoder lives in hsc_amd.hierarchical and is re-exported below.

Added (the reference has no batch dimension): ConvolutionalMatchingPursuit.computeCoefficientsBatch.

There is no CPU implementation of the greedy loop in this package: without libhscmp.so / an
MI355X the calls raise hsc_amd._native.HscmpError.
"""
import logging

import ctypes

import numpy as np
import scipy.sparse

from . import _native
from .utils import overlapAdd, centered_span

logger = logging.getLogger(__name__)


def _compute_dtype(*dtypes):
    dt = np.result_type(*dtypes)
    if dt == np.float32:
        return np.dtype(np.float32)
    return np.dtype(np.float64)     # float64, and everything else (ints, float16, ...) promotes to it


def convolve1d(sequence, filters, padding='valid', device=0):
    """hsc/modeling.py:149-188 -- cross-correlation (no flip) of `sequence` [T] or [T,F] with
    `filters` [K,W] or [K,W,F]; 'same' is zero padded with (W/2-1, W/2) for even W and
    (W//2, W//2) for odd W.  Returns [Tout, K]."""
    if padding not in ('valid', 'same'):
        raise Exception('Padding not supported: %s' % (padding))
    seq2 = np.atleast_2d(sequence).reshape((sequence.shape[0], -1))
    nbFeatures = 1 if filters.ndim == 2 else filters.shape[-1]
    assert nbFeatures == seq2.shape[-1]
    dt = _compute_dtype(seq2.dtype, filters.dtype)
    eng = _native.default_engine(device)
    eng.set_dictionary(np.asarray(filters, dtype=dt))
    return eng.convolve1d(np.asarray(seq2, dtype=dt), same=(padding == 'same'))


def convolve1d_batch(sequences, filters, padding='valid', device=0):
    """hsc/modeling.py:190-224 -- convolve1d over a leading batch axis: [B,T(,F)] -> [B,Tout,K]."""
    if padding not in ('valid', 'same'):
        raise Exception('Padding not supported: %s' % (padding))
    seqs = np.atleast_3d(sequences).reshape((sequences.shape[0], sequences.shape[1], -1))
    nbFeatures = 1 if filters.ndim == 2 else filters.shape[-1]
    assert nbFeatures == seqs.shape[-1]
    dt = _compute_dtype(seqs.dtype, filters.dtype)
    eng = _native.default_engine(device)
    eng.set_dictionary(np.asarray(filters, dtype=dt))
    return np.stack([eng.convolve1d(np.asarray(s, dtype=dt), same=(padding == 'same')) for s in seqs], axis=0)


def _host_overlap_add(signal, rows, cols, data, D3):
    """hscmp_host_overlap_add: the reference's event-by-event overlap-add in C (float64 signal and coefficients).
    Returns False when the library is not available (the numpy formulation below gives the same result)."""
    try:
        lib = _native.load_library()
    except Exception:
        return False
    Dc = np.ascontiguousarray(D3)
    r = np.ascontiguousarray(rows, dtype=np.int64); c = np.ascontiguousarray(cols, dtype=np.int64)
    v = np.ascontiguousarray(data, dtype=np.float64)
    assert signal.flags['C_CONTIGUOUS']
    rc = lib.hscmp_host_overlap_add(_native._ptr(signal), signal.shape[0], signal.shape[1], _native._ptr(r), _native._ptr(c),
                                    _native._ptr(v), len(v), _native._ptr(Dc), Dc.shape[1], 1 if Dc.dtype == np.float32 else 0)
    return rc == 0


def reconstructSignal(coefficients, D):
    """hsc/modeling.py:226-263 -- synthesis: sum of c * D[k] centred at t over the non-zero
    coefficients [T,K] (sparse or dense).  Host-side overlap-add of the (few) events."""
    assert coefficients.ndim == 1 or coefficients.ndim == 2
    assert D.ndim == 2 or D.ndim == 3
    squeezeOutput = D.ndim == 2
    D3 = D[:, :, np.newaxis] if D.ndim == 2 else D
    signal = np.zeros((coefficients.shape[0], D3.shape[-1]), dtype=coefficients.dtype)
    if scipy.sparse.issparse(coefficients):
        cx = coefficients.tocoo()
        rows, cols, data = cx.row, cx.col, cx.data
    else:
        dense = np.asarray(coefficients)
        rows, cols = np.nonzero(dense)
        data = dense[rows, cols]
    keep = data != 0.0
    rows, cols, data = rows[keep], cols[keep], data[keep]
    if signal.dtype == np.float64 and data.dtype == np.float64 and D3.dtype in (np.float32, np.float64) and len(data) > 0 \
            and _host_overlap_add(signal, rows, cols, data, D3):
        pass            # the sequential overlap-add ran in the native library (same sums, same order)
    elif np.result_type(data.dtype, D3.dtype) != signal.dtype:
        # mixed precision: keep numpy's scalar-by-scalar casting of the reference loop
        for t, k, c in zip(rows, cols, data):
            overlapAdd(signal, c * D3[k], int(t), copy=False)
    else:
        # the same sums in the same order (np.add.at is unbuffered: per sample, the events add up in
        # event order exactly as the reference's sequential overlap-add), a block of events at a time
        T, (W, Fd) = signal.shape[0], D3.shape[1:]
        flat = signal.reshape(-1)
        taps = np.arange(W, dtype=np.int64)
        feats = np.arange(Fd, dtype=np.int64)
        one_pass = signal.dtype == np.float64 and len(data) * W * Fd <= (1 << 25)
        step = len(data) if one_pass else max(1, (1 << 22) // (W * Fd))
        for i0 in range(0, len(data), max(1, step)):
            r = np.asarray(rows[i0:i0 + step], dtype=np.int64)
            pos = r[:, np.newaxis] - (W - 1) // 2 + taps[np.newaxis, :]                  # utils.py:84-99
            elems = (data[i0:i0 + step, np.newaxis, np.newaxis] * D3[cols[i0:i0 + step]]).reshape(-1, Fd)   # c * D[k]
            pos = pos.reshape(-1)
            if r.size and (int(r.min()) < (W - 1) // 2 or int(r.max()) + W // 2 >= T):
                inside = np.flatnonzero((pos >= 0) & (pos < T))                          # clipped at the edges
                pos, elems = pos[inside], elems[inside]
            idx = pos if Fd == 1 else (pos[:, np.newaxis] * Fd + feats[np.newaxis, :]).reshape(-1)
            if one_pass:
                # bincount accumulates its float64 weights one by one in input order, from 0.0
                flat[:] = np.bincount(idx, weights=elems.reshape(-1), minlength=flat.shape[0])
            else:
                np.add.at(flat, idx, elems.reshape(-1))
    if squeezeOutput:
        signal = np.squeeze(signal, axis=1)
    return signal


class Atom(object):
    """hsc/modeling.py:840-864"""

    def __init__(self, position, index, coefficient, length):
        self.__dict__.update(position=position, index=index, coefficient=coefficient, length=length)

    def getPositionSpanIndices(self, sequenceLength=None):
        startIdx = self.position - (self.length - 1) // 2
        endIdx = self.position + self.length // 2
        if sequenceLength is not None:
            startIdx = max(startIdx, 0)
            endIdx = min(endIdx, sequenceLength - 1)
        return startIdx, endIdx

    def __str__(self):
        return 'Atom no.%d of length %d at position %d, c = %4.10f' % (self.index, self.length, self.position, self.coefficient)

    __repr__ = __str__


class SparseApproximator(object):
    """hsc/modeling.py:657-660 -- the plugin seam: anything with computeCoefficients(X, D, **kw)."""

    def computeCoefficients(self, X, D):
        raise NotImplementedError()


def _slots_to_csc(slot_t, slot_k, slot_a, n, shape, minCoefficients):
    """hsc/modeling.py:1171-1181: clip |c| < minCoefficients, CSC, eliminate zeros (assembled by the native
    library: hscmp_host_slots_to_csc)."""
    lib = _native.load_library()
    n = int(n)
    t = np.ascontiguousarray(slot_t[:n], dtype=np.int32); k = np.ascontiguousarray(slot_k[:n], dtype=np.int32)
    a = np.ascontiguousarray(slot_a[:n], dtype=np.float64)
    indptr = np.empty((shape[1] + 1,), dtype=np.int32)
    indices = np.empty((max(n, 1),), dtype=np.int32); data = np.empty((max(n, 1),), dtype=np.float64)
    minc = float('nan') if minCoefficients is None else float(minCoefficients)
    rc = lib.hscmp_host_slots_to_csc(_native._ptr(t), _native._ptr(k), _native._ptr(a), n, int(shape[1]), ctypes.c_double(minc),
                                     _native._ptr(indptr), _native._ptr(indices), _native._ptr(data))
    if rc != 0:
        raise _native.HscmpError('hscmp_host_slots_to_csc failed (%d)' % rc)
    nnz = int(indptr[-1])
    m = scipy.sparse.csc_matrix((data[:nnz], indices[:nnz], indptr), shape=shape, copy=False)
    m.has_sorted_indices = True
    return m


class BatchResult(object):
    """Per-signal outputs of computeCoefficientsBatch (everything the reference returns, plus the
    ordered selection trace the reference only logs)."""

    def __init__(self, coefficients, residuals, events, stats, energies, variant, kernel_ms):
        self.coefficients = coefficients      # list of csc_matrix float64 [T,K]
        self.residuals = residuals            # [B,T] or [B,T,F]
        self.events = events                  # list of (t int32[n], k int32[n], c dtype[n]) in selection order
        self.stats = stats                    # int32 [B,8], hscmp.h HSCMP_STAT_*
        self.energies = energies              # float64 [B,2]: signal, tracked residual
        self.variant = variant
        self.kernel_ms = kernel_ms

    def stop_reasons(self):
        if self.stats is None:                # (a result of LoCOMP's host loop, which keeps no per-signal counters)
            return [None] * len(self.coefficients)
        return [_native.STOP_NAMES.get(int(s), int(s)) for s in self.stats[:, _native.STAT_STOP]]


class ConvolutionalMatchingPursuit(SparseApproximator):
    """hsc/modeling.py:866-1186, greedy convolutional matching pursuit, on the GPU."""

    def __init__(self, verbose=False, device=0):
        self.verbose = verbose
        self.device = device
        self.fig = None
        self.lastResult = None

    # ---- the reference's overridable helpers, GPU backed (LoCOMP-style subclasses build on them) ----
    def _selectBestAtoms(self, innerProducts, filterWidth, nbBlocks=1, offset=False, nullCoeffThres=0.0, weights=None):
        """hsc/modeling.py:899-982 on a materialised table [T,K]; returns the reference's list of Atom."""
        if weights is not None:
            assert len(weights) == innerProducts.shape[1]
        if isinstance(innerProducts, _native.DeviceTable):          # the table lives on the device: nothing to upload
            innerProducts.flush()                                   # (the deferred updates of the previous round)
            t, k, c = innerProducts.engine.table_select(nbBlocks, offset, nullCoeffThres, weights)
            return [Atom(int(p), int(f), cc, filterWidth) for p, f, cc in zip(t, k, c)]
        dt = _compute_dtype(innerProducts.dtype)
        eng = _native.default_engine(self.device)
        t, k, c = eng.select_best_atoms(np.asarray(innerProducts, dtype=dt), filterWidth, nbBlocks, offset, nullCoeffThres,
                                        None if weights is None else np.asarray(weights, dtype=dt))
        return [Atom(int(p), int(f), cc, filterWidth) for p, f, cc in zip(t, k, c)]

    def _updateCoefficients(self, coefficients, atoms, replace=True):
        """hsc/modeling.py:984-994 (host: a handful of scalar updates of the sparse matrix)"""
        for atom in atoms:
            if replace:
                coefficients[atom.position, atom.index] = atom.coefficient
            else:
                coefficients[atom.position, atom.index] += atom.coefficient
        return coefficients

    def _updateResidual(self, residual, energyResidual, atoms, D, eps=1e-16):
        """hsc/modeling.py:996-1016 (host: W samples per atom)"""
        energyLoss = 0.0
        for atom in atoms:
            s, e, es, ee = centered_span(residual.shape[0], atom.length, atom.position)
            before = np.sum(np.square(residual[s:e]))
            residual[s:e] += (-atom.coefficient * D[atom.index])[es:ee]
            after = np.sum(np.square(residual[s:e]))
            energyLoss += (before - after)
        if energyLoss < 0.0 and np.abs(energyLoss) > eps:
            logger.warning('Residual energy (%f) increased by %4.18f' % (energyResidual, -energyLoss))
        energyResidual -= energyLoss
        return residual, energyResidual

    def _updateInnerProducts(self, innerProducts, residual, atoms, D):
        """hsc/modeling.py:1018-1051: rows p-(W-1)..p+(W-1) re-correlated on the GPU, in place."""
        if isinstance(innerProducts, _native.DeviceTable):
            # device-resident table: hand over the residual samples the atoms changed (the union of their supports),
            # the rows around every atom are recomputed where the table lives
            T = residual.shape[0]
            spans = [centered_span(T, a.length, a.position)[:2] for a in atoms]
            lo, hi = min(s for s, _ in spans), max(e for _, e in spans)
            innerProducts.defer_update(residual, lo, hi, [a.position for a in atoms])
            # Deferring is exact for rows that read the residual as it is: whatever a later atom of the round changes inside their
            # windows it re-correlates itself.  Rows that read REFLECTED samples (:1046: a window that crosses a signal end) are not
            # covered by that argument -- a later atom may change a sample they see through the reflection without re-correlating
            # them, and the reference's table then keeps the value computed now (short signals, T < 3W - 2, are all edge).  Such
            # an update goes to the device at once.
            W = atoms[0].length if atoms else 0
            off = (W - 1) // 2
            if any(a.position - off - (W - 1) < 0 or a.position + W // 2 + (W - 1) > T - 1 for a in atoms):
                innerProducts.flush()
            return innerProducts
        dt = _compute_dtype(innerProducts.dtype, D.dtype)
        if innerProducts.dtype != dt or not innerProducts.flags.c_contiguous:
            raise TypeError('innerProducts must be a C-contiguous %s array' % dt)
        eng = _native.default_engine(self.device)
        eng.set_dictionary(np.asarray(D.reshape((D.shape[0], D.shape[1], -1)), dtype=dt))
        for atom in atoms:
            eng.update_inner_products(innerProducts, np.asarray(residual, dtype=dt), atom.position)
        return innerProducts

    # ---------------------------------------------------------------------------------------
    def computeCoefficientsBatch(self, sequences, D, nbNonzeroCoefs=None, toleranceResidualScale=None,
                                 toleranceSnr=None, nbBlocks=1, minCoefficients=1e-16, weights=None,
                                 stopCondition=None, maxEvents=None):
        """Batch form of computeCoefficients: `sequences` is [B,T] (D [K,W]) or [B,T,F] (D [K,W,F]).
        The B signals are independent (no cross-signal term in modeling.py:1053-1186) and are
        encoded concurrently, one persistent workgroup each.  Returns a BatchResult."""
        assert sequences.ndim == 2 or sequences.ndim == 3
        assert D.ndim == 2 or D.ndim == 3
        eps = float(np.finfo(D.dtype).eps) if np.issubdtype(D.dtype, np.floating) else float(np.finfo(np.float64).eps)
        dt = _compute_dtype(sequences.dtype, D.dtype)
        B, T = sequences.shape[0], sequences.shape[1]
        x = np.ascontiguousarray(sequences.reshape((B, T, -1)), dtype=dt)
        D3 = np.ascontiguousarray(D.reshape((D.shape[0], D.shape[1], -1)), dtype=dt)
        K, W, F = D3.shape
        assert F == x.shape[2]
        if weights is not None:
            assert len(weights) == K
        eng = _native.engine_for(self.device, D3, None if weights is None else np.asarray(weights, dtype=dt))

        if maxEvents is None:
            maxEvents = 2 * int(nbNonzeroCoefs) + 64 if nbNonzeroCoefs is not None else 4096
        per_round = stopCondition is not None
        params = _native.make_params(nbNonzeroCoefs, toleranceResidualScale, toleranceSnr, nbBlocks,
                                     minCoefficients, eps, maxEvents, 1 if per_round else 0)
        method = getattr(self, '_method', _native.METHOD_CMP)       # (hsc_amd.locomp.LoCOMP: the loop with the group re-fit)
        if method != _native.METHOD_CMP:
            eng.set_method(method)
        try:
            eng.encode_batch(x, params)
        finally:
            if method != _native.METHOD_CMP:
                eng.set_method(_native.METHOD_CMP)                   # (engines are shared; a resumed batch keeps its own loop)
        kernel_ms = list(eng.last_kernel_ms())
        while True:
            if per_round:
                self._run_with_callback(eng, sequences, D, K, T, minCoefficients, stopCondition)
            stats = eng.fetch_stats()
            if not np.any(stats[:, _native.STAT_STOP] == _native.STOP_CAPACITY):
                break
            # the event list was too short for this stop rule: enlarge it and resume where the loop stopped
            # (a round is only started when all its atoms fit, so the trace equals an uninterrupted run)
            bound = _native.max_event_capacity(T)
            if maxEvents >= bound:
                raise _native.HscmpError('the pursuit does not converge: more than %d selections per signal without meeting a stop '
                                         'rule (the same atoms are re-selected; the reference would not terminate)' % maxEvents)
            maxEvents = min(4 * maxEvents, bound)          # enlarge the event lists and resume (exact, see hscmp_grow_events)
            eng.grow_events(maxEvents)
            eng.continue_rounds(1 if per_round else 0)
            kernel_ms[2] += eng.last_kernel_ms()[2]

        ev_t, ev_k, ev_c = eng.fetch_events()
        st, sk, sa = eng.fetch_slots()
        residuals = eng.fetch_residual()
        energies = eng.fetch_energies()
        coefficients, events = [], []
        for b in range(B):
            ne, ns = int(stats[b, _native.STAT_EVENTS]), int(stats[b, _native.STAT_SLOTS])
            coefficients.append(_slots_to_csc(st[b], sk[b], sa[b], ns, (T, K), minCoefficients))
            events.append((ev_t[b, :ne].copy(), ev_k[b, :ne].copy(), ev_c[b, :ne].copy()))
        if sequences.ndim == 2 or D.ndim == 2:
            residuals = np.squeeze(residuals, axis=2)                      # modeling.py:1183-1184
        if residuals.dtype != sequences.dtype and np.issubdtype(sequences.dtype, np.floating):
            residuals = residuals.astype(sequences.dtype)
        res = BatchResult(coefficients, residuals, events, stats, energies, eng.last_variant(), kernel_ms)
        self.lastResult = res
        if self.verbose:
            for b in range(B):
                for t, k, c in zip(*events[b]):
                    logger.info('Matching pursuit: raw event is (t = %d, f = %d, c = %f)' % (t, k, c))
        for b in range(B):
            logger.debug('signal %d: %d selections in %d rounds, nnz %d, duplicates %d, stop: %s' % (
                b, stats[b, _native.STAT_ITERATIONS], stats[b, _native.STAT_ROUNDS], stats[b, _native.STAT_NNZ],
                stats[b, _native.STAT_DUPLICATES], _native.STOP_NAMES.get(int(stats[b, _native.STAT_STOP]))))
        return res

    def _run_with_callback(self, eng, sequences, D, K, T, minCoefficients, stopCondition):
        """modeling.py:1155-1158: stopCondition(sequence, residual, coefficients) after every
        selection round.  One GPU launch per round; the callback sees host copies."""
        while True:
            stats = eng.fetch_stats()
            running = np.where(stats[:, _native.STAT_STOP] == _native.STOP_RUNNING)[0]
            if running.size == 0:
                return
            st, sk, sa = eng.fetch_slots()
            residuals = eng.fetch_residual()
            for b in running:
                ns = int(stats[b, _native.STAT_SLOTS])
                coef = scipy.sparse.lil_matrix((T, K))
                if ns > 0:
                    coef[st[b, :ns], sk[b, :ns]] = sa[b, :ns]
                seq = sequences[b].reshape((T, -1))                        # the reference passes [T,F] views
                if stopCondition(seq, residuals[b], coef):
                    logger.warning('Custom stop condition reached: considering convergence is achieved')
                    eng.stop_signal(int(b))
            eng.continue_rounds(1)

    # ---------------------------------------------------------------------------------------
    def computeCoefficients(self, sequence, D, nbNonzeroCoefs=None, toleranceResidualScale=None, toleranceSnr=None,
                            nbBlocks=1, minCoefficients=1e-16, weights=None, stopCondition=None):
        """hsc/modeling.py:1053-1186.  Returns (scipy.sparse.csc_matrix float64 [T,K], residual
        with the shape / dtype rules of :1183-1186)."""
        assert sequence.ndim == 1 or sequence.ndim == 2
        assert D.ndim == 2 or D.ndim == 3
        sequence = np.asarray(sequence)
        res = self.computeCoefficientsBatch(sequence[np.newaxis], D, nbNonzeroCoefs, toleranceResidualScale,
                                            toleranceSnr, nbBlocks, minCoefficients, weights, stopCondition)
        return res.coefficients[0], res.residuals[0]


class ConvolutionalSparseCoder(object):
    """hsc/modeling.py:1656-1669"""

    def __init__(self, D, approximator):
        assert D.ndim == 2 or D.ndim == 3
        self.D = D
        self.approximator = approximator

    def encode(self, X, *args, **kwargs):
        assert X.ndim == 1 or X.ndim == 2
        return self.approximator.computeCoefficients(X, self.D, *args, **kwargs)

    def encodeBatch(self, X, *args, **kwargs):
        return self.approximator.computeCoefficientsBatch(X, self.D, *args, **kwargs)

    def reconstruct(self, coefficients):
        assert coefficients.ndim == 1 or coefficients.ndim == 2
        return reconstructSignal(coefficients, self.D)


# The hierarchical encoder (hsc/modeling.py:1427-1705) lives in hsc_amd.hierarchical and is reachable
# from here under the reference's names (lazy, PEP 562: hierarchical itself imports this module).
def __getattr__(name):
    if name in ('HierarchicalConvolutionalMatchingPursuit', 'HierarchicalConvolutionalSparseCoder'):
        from . import hierarchical
        return getattr(hierarchical, name)
    if name == 'LoCOMP':
        from . import locomp
        return locomp.LoCOMP
    if name in ('ConvolutionalDictionaryLearner', 'extractRandomWindows', 'extractWindows', 'extractWindowsBatch'):
        from . import learning
        return getattr(learning, name)
    raise AttributeError('module %r has no attribute %r' % (__name__, name))
