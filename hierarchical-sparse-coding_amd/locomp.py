"""LoCOMP -- low-complexity orthogonal matching pursuit (reference: hsc/modeling.py:1191-1425;
Mailhe et al., "A low complexity Orthogonal Matching Pursuit for sparse signal approximation with
shift-invariant dictionaries", ICASSP 2009).

Same outer loop as the greedy coder, but after each selection the coefficients of all previously
selected atoms whose support overlaps the new one are re-fitted jointly (least squares on the local
residual), then the residual and the inner products are updated for every atom of that group.

The heavy operations run on the GPU through the same C ABI as the greedy coder.  The materialised table
`innerProducts` [T, K] and a copy of the residual live ON THE DEVICE for the whole encode (hscmp_table_open =
the initial correlation, hscmp_table_select = the selection, hscmp_table_update = local re-correlation in
place): per iteration the host sends the ~3W residual samples a re-fitted group changed and receives the
selected atoms.  The loop itself and the tiny least-squares systems (a handful of atoms by ~3W samples,
np.linalg.pinv as in the reference, :1326) are host side.  The table-free persistent-kernel engine is
ConvolutionalMatchingPursuit.
"""
import bisect
import logging
import os

import numpy as np
import scipy.sparse

from . import _native
from .modeling import Atom, ConvolutionalMatchingPursuit, _compute_dtype
from .utils import overlapAdd, peek

logger = logging.getLogger(__name__)


class LoCOMP(ConvolutionalMatchingPursuit):

    def __init__(self, verbose=False, device=0, refit='device'):
        """refit='device': the loop runs inside the greedy-loop kernel (csrc/hscmp_locomp.h), the group re-fit as float64 normal
        equations.  refit='host': the host loop below over the device-resident table, the re-fit through np.linalg.pinv in the
        dictionary's dtype exactly as the reference (:1326) -- what the per-signal hierarchical entry uses: a cascade of levels
        amplifies the last-bit differences between two solvers wherever a group is ill-conditioned."""
        super(LoCOMP, self).__init__(verbose, device)
        assert refit in ('device', 'host')
        self.refit = refit

    def _initialInnerProducts(self, residual, D, dt):
        """:1293 innerProducts = convolve1d(residual, D, padding='same') -- computed and KEPT on the device
        (hscmp_table_open): _selectBestAtoms reads it there, _updateInnerProducts edits it there.  Returns the handle
        (_native.DeviceTable) the two hooks understand."""
        eng = _native.engine_for(self.device, np.asarray(D, dtype=dt))
        return eng.table_open(np.asarray(residual, dtype=dt))

    def _findCommonSupportAtoms(self, atom, coefficients, D):
        """Previously selected atoms in the neighbourhood of `atom` (:1222-1241).  As in the reference
        the exclusion test compares the window-relative row with the absolute position, and drops
        every entry that shares the atom's index."""
        T, W = coefficients.shape[0], D.shape[1]
        start, end = atom.getPositionSpanIndices(T)
        start = max(start - W // 2, 0)
        end = min(end + (W // 2 - 1 if W % 2 == 0 else W // 2), T)
        index = getattr(self, '_support', None)
        if index is not None and index[0] is coefficients:
            # the encode loop keeps the non-zero entries by position next to the matrix: same entries, same order
            # (rows ascending, columns ascending) as slicing the T x K list-of-lists matrix, without building one
            _, positions, entries = index
            out = []
            for q in range(bisect.bisect_left(positions, start), bisect.bisect_right(positions, end)):
                t = positions[q]
                for k in sorted(entries[t]):
                    if (t - start) != atom.position and k != atom.index:
                        out.append(Atom(t, k, entries[t][k], W))
            return out
        sub = coefficients[start:end + 1, :].tocoo()
        return [Atom(start + int(r), int(k), c, W) for r, k, c in zip(sub.row, sub.col, sub.data)
                if r != atom.position and k != atom.index]

    def _track(self, coefficients, atoms):
        """Mirror of the matrix entries the atoms touched (a list-of-lists matrix drops an entry that became 0.0)."""
        _, positions, entries = self._support
        for a in atoms:
            v = coefficients[a.position, a.index]
            row = entries.get(a.position)
            if v != 0.0:
                if row is None:
                    row = entries[a.position] = {}
                    bisect.insort(positions, a.position)
                row[a.index] = v
            elif row is not None and a.index in row:
                del row[a.index]
                if not row:
                    del entries[a.position]
                    positions.pop(bisect.bisect_left(positions, a.position))

    def _getDictionaryFromSupportAtoms(self, sequence, atoms, D):
        """Local dictionary of the group: every atom placed on the union of the supports (:1243-1265)."""
        lo = min(a.getPositionSpanIndices(sequence.shape[0])[0] for a in atoms)
        hi = max(a.getPositionSpanIndices(sequence.shape[0])[1] for a in atoms)
        Dsup = []
        for a in atoms:
            s = np.zeros((hi - lo + 1,) + sequence.shape[1:], dtype=D.dtype)
            overlapAdd(s, D[a.index], a.position - lo, copy=False)
            Dsup.append(s)
        return np.stack(Dsup), sequence[lo:hi + 1]

    _method = _native.METHOD_LOCOMP      # computeCoefficientsBatch of the base class then runs the device loop of hscmp_locomp.h

    def computeCoefficientsBatch(self, sequences, D, nbNonzeroCoefs=None, toleranceResidualScale=None, toleranceSnr=None, nbBlocks=1,
                                 minCoefficients=1e-16, weights=None, stopCondition=None, maxEvents=None):
        """The B signals of `sequences` side by side, one workgroup each, through the LoCOMP atom body of the greedy-loop kernel
        (csrc/hscmp_locomp.h: neighbourhood from the device slot list, normal equations of the group in LDS, group update and
        re-correlation in the kernel).  A signal with a neighbourhood beyond the kernel's capacity (stop reason 'group') is
        taken again by the host loop below; so is every call with a stopCondition (its argument there is the coefficient matrix,
        :1397) or with verbose plots.  Returns a BatchResult like the base class."""
        if stopCondition is not None or self.verbose or self.refit == 'host' or os.environ.get('HSCMP_LOCOMP_HOST') == '1':
            return self._batch_on_host(sequences, D, nbNonzeroCoefs, toleranceResidualScale, toleranceSnr, nbBlocks, minCoefficients, weights,
                                       stopCondition)
        res = super(LoCOMP, self).computeCoefficientsBatch(sequences, D, nbNonzeroCoefs, toleranceResidualScale, toleranceSnr, nbBlocks,
                                                           minCoefficients, weights, None, maxEvents)
        again = np.where(res.stats[:, _native.STAT_STOP] == _native.STOP_GROUP)[0]
        for b in again:
            coef, residual = self._computeCoefficientsHost(np.asarray(sequences[b]), D, nbNonzeroCoefs, toleranceResidualScale, toleranceSnr,
                                                           nbBlocks, minCoefficients, weights, None)
            res.coefficients[b] = coef
            res.residuals[b] = residual
            # what describes the signal now is the host loop's run, not the device run that was given up: counters the host loop
            # can answer are refreshed, the others cleared, the stop reason says who finished the signal
            if res.stats is not None:
                res.stats[b, :] = 0
                res.stats[b, _native.STAT_NNZ] = coef.nnz
                res.stats[b, _native.STAT_STOP] = _native.STOP_HOST
            if res.energies is not None:
                res.energies[b, 1] = float(np.sum(np.square(np.asarray(residual, dtype=np.float64))))
            if res.events is not None:
                res.events[b] = (np.zeros((0,), np.int32), np.zeros((0,), np.int32), np.zeros((0,), res.events[b][2].dtype))
        return res

    def _batch_on_host(self, sequences, D, *args):
        from .modeling import BatchResult
        out = [self._computeCoefficientsHost(np.asarray(s), D, *args) for s in sequences]
        res = BatchResult([o[0] for o in out], np.stack([o[1] for o in out], axis=0), None, None, None, 'locomp_host', None)
        self.lastResult = res                 # (no event trace, counters or stop reasons: the host loop keeps none)
        return res

    def computeCoefficients(self, sequence, D, nbNonzeroCoefs=None, toleranceResidualScale=None, toleranceSnr=None,
                            nbBlocks=1, minCoefficients=1e-16, weights=None, stopCondition=None):
        """hsc/modeling.py:1267-1425"""
        res = self.computeCoefficientsBatch(np.asarray(sequence)[np.newaxis], D, nbNonzeroCoefs, toleranceResidualScale, toleranceSnr, nbBlocks,
                                            minCoefficients, weights, stopCondition)
        return res.coefficients[0], res.residuals[0]

    def _computeCoefficientsHost(self, sequence, D, nbNonzeroCoefs=None, toleranceResidualScale=None, toleranceSnr=None,
                                 nbBlocks=1, minCoefficients=1e-16, weights=None, stopCondition=None):
        """hsc/modeling.py:1267-1425 as a host loop over the table entry points (hscmp_table_open / _select / _update)"""
        assert sequence.ndim == 1 or sequence.ndim == 2
        assert D.ndim == 2 or D.ndim == 3
        eps = np.finfo(D.dtype).eps
        squeezeOutput = sequence.ndim == 1 or D.ndim == 2
        if sequence.ndim == 1:
            sequence = sequence[:, np.newaxis]
        if D.ndim == 2:
            D = D[:, :, np.newaxis]
        dt = _compute_dtype(sequence.dtype, D.dtype)

        energySignal = np.sum(np.square(sequence))
        residual = np.copy(sequence)
        energyResidual = energySignal
        # (a dictionary-of-keys matrix: the same `coefficients[t, k] += c` / `.nnz` / `.tocoo()` behaviour as the reference's
        # list-of-lists matrix, without the T empty row lists -- 17 ms to build at T = 65536)
        coefficients = scipy.sparse.dok_matrix((sequence.shape[0], D.shape[0]), dtype=np.float64)
        self._support = (coefficients, [], {})
        innerProducts = self._initialInnerProducts(residual, D, dt)                                                   # :1293

        offset = False
        converged = False
        while not converged:
            atoms = self._selectBestAtoms(innerProducts, nbBlocks=nbBlocks, filterWidth=D.shape[1], offset=offset,
                                          nullCoeffThres=minCoefficients, weights=weights)
            if toleranceSnr is not None and len(atoms) > 1:                       # weak-atom filter :1303-1312
                limit = energySignal / (10.0 ** (toleranceSnr / 10.0)) / np.prod(sequence.shape)
                atoms = [a for a in atoms if np.mean(np.square(peek(residual, a.length, a.position))) >= limit]

            for atom in atoms:
                if coefficients[atom.position, atom.index] != 0.0:
                    logger.warning('Redundant atom selected: %s' % (str(atom)))
                lastEnergyResidual = energyResidual
                group = self._findCommonSupportAtoms(atom, coefficients, D)
                if len(group) > 0:
                    # joint least-squares re-fit of the group on the local residual (:1322-1341)
                    group = [atom] + group
                    Dsup, residualSup = self._getDictionaryFromSupportAtoms(residual, group, D)
                    DsupF = Dsup.reshape((Dsup.shape[0], -1))
                    fitted = np.dot(np.linalg.pinv(DsupF).T, residualSup.flatten()).flatten()
                    for a, c in zip(group, fitted):
                        a.coefficient = c
                else:
                    group = [atom]
                coefficients = self._updateCoefficients(coefficients, group, replace=False)
                self._track(coefficients, group)
                residual, energyResidual = self._updateResidual(residual, energyResidual, group, D, eps)
                innerProducts = self._updateInnerProducts(innerProducts, residual, group, D)

                if energyResidual < eps:                                          # :1360-1365
                    converged = True
                    break
                snr = 10.0 * np.log10(energySignal / energyResidual)
                if nbNonzeroCoefs is not None and coefficients.nnz >= nbNonzeroCoefs:
                    converged = True
                    break
                if toleranceSnr is not None and snr >= toleranceSnr:
                    converged = True
                    break
                if np.abs(lastEnergyResidual - energyResidual) < eps:             # :1379-1383
                    logger.warning('Residual energy is no more reduced: considering convergence is achieved')
                    converged = True
                    break

            if toleranceResidualScale is not None and np.max(np.abs(residual)) <= toleranceResidualScale:
                converged = True
            if len(atoms) == 0:
                logger.warning('Selection returned empty set: considering convergence is achieved')
                converged = True
            if stopCondition is not None and stopCondition(coefficients):         # single-argument form, :1397
                converged = True
            offset = not offset

        self._support = None
        if minCoefficients is not None:                                           # :1411-1417
            cx = coefficients.tocoo()
            keep = np.abs(cx.data) >= minCoefficients
            coefficients = scipy.sparse.coo_matrix((cx.data[keep], (cx.row[keep], cx.col[keep])), shape=cx.shape)
        coefficients = scipy.sparse.csc_matrix(coefficients)
        coefficients.sum_duplicates()
        coefficients.eliminate_zeros()
        if squeezeOutput:
            residual = np.squeeze(residual, axis=1)
        return coefficients, residual
