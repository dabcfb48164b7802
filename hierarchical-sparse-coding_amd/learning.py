"""Dictionary learners that drive the GPU engine (reference: hsc/modeling.py:82-147, 265-655).

SURVEY.md section 8 f-4: the convolutional k-means learner (Dundar et al., ICLR 2016) spends its time
in `convolve1d_batch(windows, D, 'valid')` + a per-window arg-max (modeling.py:454-460) -- the same
correlate -> arg-max pattern as the matching pursuit, over a batch of 2W-sample windows.  That step
runs on the GPU (hscmp_assign_windows); window extraction, the centroid update and the resets stay on
the host and draw from numpy's RandomState in the reference's order, so a seeded run reproduces the
reference's dictionary.  K-SVD (modeling.py:526-641) is provided on top of the GPU sparse coders.
The multiplicative NMF learner (:330-417) is a different algorithm and out of scope.
"""
import logging

import numpy as np
import scipy.linalg

from . import _native
from .utils import normalize

logger = logging.getLogger(__name__)


def _rng(rng):
    return np.random if rng is None else rng


def extractWindows(sequence, indices, width, centered=False):
    """Windows sequence[i : i+width] (or centred at i) for every i of `indices` (hsc/modeling.py:91-117)."""
    assert sequence.ndim == 1 or sequence.ndim == 2
    assert width > 0 and width < sequence.shape[0]
    seq = sequence[:, np.newaxis] if sequence.ndim == 1 else sequence
    starts = np.asarray(indices, dtype=np.int64)
    if centered:
        starts = starts - (width - 1) // 2
    starts = np.where(starts < 0, starts + (seq.shape[0] - width + 1), starts)      # numpy's negative-index wrap of the strided view
    windows = seq[starts[:, np.newaxis] + np.arange(width)[np.newaxis, :]]
    return windows[:, :, 0] if sequence.ndim == 1 else windows


def extractRandomWindows(sequence, nbWindows, width, rng=None):
    """hsc/modeling.py:82-89"""
    assert sequence.ndim == 1 or sequence.ndim == 2
    assert nbWindows > 0
    assert width > 0 and width < sequence.shape[0]
    indices = _rng(rng).randint(low=0, high=sequence.shape[0] - width, size=(nbWindows,))
    return extractWindows(sequence, indices, width, centered=False)


def extractWindowsBatch(sequences, indices, width, centered=False):
    """One window per sequence of the batch [N, L(, F)] (hsc/modeling.py:119-147)."""
    assert sequences.ndim == 2 or sequences.ndim == 3
    assert width > 0 and width <= sequences.shape[1]
    seqs = sequences[:, :, np.newaxis] if sequences.ndim == 2 else sequences
    starts = np.asarray(indices, dtype=np.int64)
    if centered:
        starts = starts - (width - 1) // 2
    starts = np.where(starts < 0, starts + (seqs.shape[1] - width + 1), starts)
    rows = np.arange(seqs.shape[0])[:, np.newaxis]
    windows = seqs[rows, starts[:, np.newaxis] + np.arange(width)[np.newaxis, :]]
    return windows[:, :, 0] if sequences.ndim == 2 else windows


class ConvolutionalDictionaryLearner(object):
    """hsc/modeling.py:265-655 (algorithms 'samples', 'kmean', 'ksvd')."""

    def __init__(self, k, windowSize, algorithm='kmean', verbose=False, device=0, rng=None):
        self.k = int(k)
        self.windowSize = int(windowSize)
        self.algorithm = algorithm
        self.verbose = verbose           # (the reference plots the centroids; here: debug logging only)
        self.device = device
        self.rng = rng
        self.fig = None

    # ---- the GPU step ----------------------------------------------------------------------------
    def _assign(self, windows, D):
        """Best (position, centroid) of every window: arg-max of |'valid' correlation| (modeling.py:454-460)."""
        from .modeling import _compute_dtype
        dt = _compute_dtype(windows.dtype, D.dtype)
        eng = _native.default_engine(self.device)
        eng.set_dictionary(np.ascontiguousarray(D.reshape((D.shape[0], D.shape[1], -1)), dtype=dt))
        t, k, _ = eng.assign_windows(np.asarray(windows, dtype=dt))
        return t.astype(np.int64), k.astype(np.int64)

    # ---- host side -------------------------------------------------------------------------------
    def _init_D(self, data, initMethod='random_samples'):
        """hsc/modeling.py:308-328"""
        assert data.ndim == 1 or data.ndim == 2
        seq = data[:, np.newaxis] if data.ndim == 1 else data
        if initMethod == 'noise':
            D = normalize(_rng(self.rng).uniform(low=np.min(seq), high=np.max(seq), size=(self.k, self.windowSize, seq.shape[-1])))
        elif initMethod == 'random_samples':
            D = normalize(extractRandomWindows(seq, self.k, self.windowSize, self.rng))
        else:
            raise Exception('Unsupported initialization method: %s' % (initMethod))
        return np.squeeze(D, axis=2) if data.ndim == 1 else D

    def _train_samples(self, data, avoidSingletons=False):
        """k random windows of the data with non-zero norm (and, optionally, more than one non-zero
        sample), normalised (hsc/modeling.py:279-306)."""
        patterns = []
        while len(patterns) < self.k:
            windows = extractRandomWindows(data, self.k, self.windowSize, self.rng)
            axes = tuple(range(1, windows.ndim))
            valid = np.sqrt(np.sum(np.square(windows), axis=axes)) > 0.0
            if avoidSingletons:
                valid &= np.sum(windows != 0.0, axis=axes) > 1
            patterns.extend(list(windows[valid])[:self.k - len(patterns)])
        return normalize(np.stack(patterns))

    def _train_kmean(self, data, nbRandomWindows, maxIterations=100, tolerance=0.0, initMethod='random_samples',
                     resetMethod='noise', nbAveragedPatches=8):
        """Convolutional k-means (hsc/modeling.py:420-524): windows twice as long as the centroids; each
        iteration assigns every window, at its best-matching offset, to its best centroid (GPU), then
        every centroid becomes the mean of its normalised patches (host)."""
        rng = _rng(self.rng)
        W = self.windowSize
        windows = extractRandomWindows(data, nbRandomWindows, 2 * W, self.rng)
        D = self._init_D(data, initMethod)
        n, alpha = 0, tolerance + 1.0
        while n < maxIterations and alpha > tolerance:
            positions, assignments = self._assign(windows, D)
            # the matched patch of every window: samples positions .. positions+W-1 (:462-470)
            patches = extractWindowsBatch(windows, (W - 1) // 2 + positions, width=W, centered=True)
            assert np.max(assignments) < D.shape[0]
            centroids, nbResets = [], 0
            for c in range(D.shape[0]):
                members = np.where(assignments == c)
                # NB the reference tests np.any() of the member INDICES (:479-480), so a centroid whose only
                # member is window 0 counts as empty; kept, it decides which random numbers are drawn
                if np.any(members):
                    centroid = np.mean(normalize(patches[members]), axis=0)          # cosine mean
                else:
                    nbResets += 1
                    if resetMethod == 'random_samples':
                        centroid = patches[rng.randint(low=0, high=patches.shape[0])]
                    elif resetMethod == 'random_samples_average':
                        centroid = np.mean(patches[rng.randint(low=0, high=patches.shape[0], size=(nbAveragedPatches,))], axis=0)
                    elif resetMethod == 'noise':
                        centroid = rng.uniform(low=-1.0, high=1.0, size=patches.shape[1:])
                    else:
                        raise Exception('Unsupported reset method: %s' % (resetMethod))
                if np.sqrt(np.sum(np.square(centroid))) == 0.0:
                    centroid = centroid + 1e-9
                centroids.append(centroid)
            newD = normalize(np.stack(centroids))
            alpha = np.sqrt(np.sum(np.square(D - newD)))
            logger.debug('K-mean iteration %d: tolerance = %f, nb resets = %d' % (n, alpha, nbResets))
            D = newD
            n += 1
        return D

    def _train_ksvd(self, data, method='locomp', maxIterations=100, tolerance=0.0, nbNonzeroCoefs=None, toleranceSnr=40.0, usePCA=False):
        """Convolutional K-SVD (hsc/modeling.py:526-641): sparse-code the data with the current dictionary
        (GPU matching pursuit), then refit every atom to the rank-1 approximation of the patches it explains."""
        from .modeling import ConvolutionalMatchingPursuit, ConvolutionalSparseCoder, LoCOMP, reconstructSignal
        if usePCA:
            raise NotImplementedError('usePCA=True needs the reference\'s pca helper, which is outside this path')
        D = self._init_D(data, initMethod='noise')
        W = D.shape[1]
        n, alpha = 0, tolerance + 1.0
        while n < maxIterations and alpha > tolerance:
            if method == 'locomp':
                coder = LoCOMP()
            elif method == 'cmp':
                coder = ConvolutionalMatchingPursuit()
            else:
                raise Exception('Unsupported sparse coding method: %s' % (method))
            coefficients, _ = ConvolutionalSparseCoder(D, coder).encode(data, nbNonzeroCoefs=nbNonzeroCoefs, toleranceSnr=toleranceSnr)
            coefficients = coefficients.tolil()
            oldD = np.copy(D)
            for k in range(D.shape[0]):
                indices = coefficients[:, k].nonzero()[0]
                if len(indices) == 0:
                    continue
                coefficients[indices, k * np.ones_like(indices)] = 0.0
                error = reconstructSignal(coefficients.tocsc(), D)          # the signal explained WITHOUT atom k (:592-597)
                padded = np.pad(error, [(W // 2, W // 2)] + [(0, 0)] * (error.ndim - 1), mode='constant')
                patches = extractWindows(padded, W // 2 + indices, width=W, centered=True)
                patches = patches.reshape((patches.shape[0], -1))
                U, s, Vh = scipy.linalg.svd(patches.T, full_matrices=False)
                D[k, :] = U[:, 0].reshape(D.shape[1:])
                coefficients[indices, k * np.ones_like(indices)] = Vh.T[:, 0] * s[0]
            alpha = np.sqrt(np.sum(np.square(D - oldD)))
            logger.debug('K-SVD iteration %d: tolerance = %f, sparsity = %f' % (n, alpha, float(coefficients.nnz) / np.prod(coefficients.shape)))
            n += 1
        return D

    def train(self, X, *args, **kwargs):
        """hsc/modeling.py:643-655"""
        if self.algorithm == 'samples':
            return self._train_samples(X, *args, **kwargs)
        if self.algorithm == 'kmean':
            return self._train_kmean(X, *args, **kwargs)
        if self.algorithm == 'ksvd':
            return self._train_ksvd(X, *args, **kwargs)
        if self.algorithm == 'nmf':
            raise NotImplementedError("algorithm='nmf' (multiplicative NMF, hsc/modeling.py:330-417) is not part of the matching-pursuit path")
        raise Exception('Unknown training algorithm: %s' % (self.algorithm))
