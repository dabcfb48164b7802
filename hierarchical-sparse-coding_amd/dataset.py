"""Host-side data model the hierarchical encoder reads (reference: hsc/dataset.py:110-410, 826-869).

Only the container and the two helpers the encoder depends on are provided -- raw dictionaries per
level, their input-level representations, singleton ("pass-through") bases.  Dictionary *generation*
(Perlin atoms, random compositions, hsc/dataset.py:412-676) and the signal generators are out of
scope of the matching-pursuit hot path.
"""
import logging
import pickle

import numpy as np

from .utils import overlapAdd

logger = logging.getLogger(__name__)


def scalesToWindowSizes(scales):
    """Filter width of each level from the input-level scales (hsc/dataset.py:862-869):
    level 0 keeps its scale, level l spans scale[l] - scale[l-1] + 1 steps of level l-1."""
    assert len(scales) > 0
    sizes = [int(scales[0])]
    for level in range(1, len(scales)):
        sizes.append(int(scales[level] - scales[level - 1] + 1))
    return np.array(sizes, dtype=int)


def addSingletonBases(dictionaries):
    """Prepend, at every level >= 1, one unit "singleton" atom per input feature so that lower-level
    events can pass through unchanged (hsc/dataset.py:826-860)."""
    assert len(dictionaries) > 1
    out = [dictionaries[0]]
    counts = [dictionaries[0].shape[0]]
    for level in range(1, len(dictionaries)):
        D = dictionaries[level]
        if level > 1:
            # the previous level grew by its own singletons: widen the feature axis (zeros in front)
            D = np.pad(D, [(0, 0), (0, 0), (out[level - 1].shape[0] - D.shape[-1], 0)], mode='constant')
        assert D.shape[-1] == out[level - 1].shape[0]
        n = counts[level - 1]
        singles = np.zeros((n,) + D.shape[1:], dtype=D.dtype)
        idx = np.arange(n)
        singles[idx, (D.shape[1] - 1) // 2, idx] = 1.0           # centre tap of the level's window
        newD = np.concatenate((singles, D), axis=0)
        out.append(newD)
        counts.append(newD.shape[0])
    return out


EVENT_DTYPE = np.dtype('int32,int32,int32,float32')      # (time, level, atom index, coefficient)


def convertSparseMatricesToEvents(coefficients):
    """Per-level coefficient matrices -> the reference's event record array, sorted by time with the
    reference's tie order (stable w.r.t. level, then COO order) (hsc/dataset.py:798-811)."""
    times, levels, indices, values = [], [], [], []
    for level, c in enumerate(coefficients):
        c = c.tocoo()
        times.append(np.asarray(c.row, dtype=np.int64))
        levels.append(np.full(c.row.shape, level, dtype=np.int64))
        indices.append(np.asarray(c.col, dtype=np.int64))
        values.append(np.asarray(c.data))
    if not times:
        return np.zeros((0,), dtype=EVENT_DTYPE)
    t = np.concatenate(times); l = np.concatenate(levels); i = np.concatenate(indices); v = np.concatenate(values)
    order = np.argsort(t, kind='stable')                  # Python's sorted() is stable
    events = np.zeros((len(t),), dtype=EVENT_DTYPE)
    events['f0'] = t[order]; events['f1'] = l[order]; events['f2'] = i[order]; events['f3'] = v[order]
    return events


def convertEventsToSparseMatrices(events, counts, sequenceLength):
    """Event records -> one CSR matrix [sequenceLength, count] per level (hsc/dataset.py:813-824)."""
    import scipy.sparse
    t = np.asarray(events['f0'], dtype=np.int64)
    l = np.asarray(events['f1'], dtype=np.int64)
    i = np.asarray(events['f2'], dtype=np.int64)
    v = np.asarray(events['f3'])
    out = []
    for level, count in enumerate(counts):
        m = l == level
        out.append(scipy.sparse.coo_matrix((v[m], (t[m], i[m])), shape=(sequenceLength, count)).tocsr())
    return out


class MultilevelDictionary(object):
    """hsc/dataset.py:110-410 (container part)."""

    def __init__(self, dictionaries, scales, representations, decompositions, hasSingletonBases=False):
        assert len(dictionaries) > 0
        assert len(scales) > 0
        self.dictionaries = dictionaries
        self.scales = scales
        self.representations = representations
        self.decompositions = decompositions
        self.hasSingletonBases = hasSingletonBases
        if decompositions is not None:
            self.counts = np.array([dictionaries[0].shape[0]] + [len(d) for d in decompositions], dtype=int)
        else:
            self.counts = np.array([d.shape[0] for d in dictionaries], dtype=int)
        if self.hasSingletonBases:
            noSingle = [self.counts[0]]
            for level in range(1, len(self.counts)):
                noSingle.append(self.counts[level] - np.sum(noSingle))
            self.countsNoSingletons = np.array(noSingle, dtype=int)
        else:
            self.countsNoSingletons = np.copy(self.counts)

    @classmethod
    def fromRawDictionaries(cls, dictionaries, scales, hasSingletonBases=False):
        """Decompositions and input-level representations from the raw per-level dictionaries
        (hsc/dataset.py:137-194)."""
        assert len(dictionaries) > 0
        assert len(scales) > 0
        widths = scalesToWindowSizes(scales)
        decompositions = []
        representations = [dictionaries[0]]
        for level, dictionary in enumerate(dictionaries):
            assert dictionary.shape[1] == widths[level]
            if level == 0:
                continue
            levelDec, levelRep = [], []
            lead = (int(scales[level - 1]) - 1) // 2        # level-relative -> input-level position
            for pattern in dictionary:
                rows, cols = np.nonzero(np.abs(pattern) > 0.0)
                if len(rows) > 0:
                    coefs = pattern[rows, cols]
                    positions = rows + lead
                    levels = (level - 1) * np.ones_like(positions, dtype=np.int32)
                    levelDec.append([levels, cols, positions, coefs])
                    signal = np.zeros(int(scales[level]), dtype=dictionary.dtype)
                    for l, i, t, c in zip(levels, cols, positions, coefs):
                        overlapAdd(signal, element=c * representations[l][i, :], t=int(t), copy=False)
                    norm = np.sqrt(np.sum(np.square(signal)))
                    assert norm > 0.0
                    levelRep.append(signal / norm)
                else:
                    levelDec.append([])
                    levelRep.append(np.zeros(int(scales[level]), dtype=dictionary.dtype))
                    logger.warning('Null pattern found in dictionary at level %d' % (level))
            decompositions.append(levelDec)
            representations.append(np.stack(levelRep))
        return cls(dictionaries, scales, representations, decompositions, hasSingletonBases)

    @classmethod
    def fromBaseDictionary(cls, baseDict):
        assert len(baseDict) > 0
        return cls(dictionaries=[baseDict], scales=[baseDict.shape[1]], representations=[baseDict], decompositions=None)

    def withSingletonBases(self):
        """hsc/dataset.py:317-331"""
        if self.hasSingletonBases:
            logger.warning('Could not add expanded bases since they already exist')
            return self
        if self.getNbLevels() <= 1:
            logger.warning('Could not add expanded bases since there is only one level')
            return self
        return MultilevelDictionary.fromRawDictionaries(addSingletonBases(self.dictionaries), self.scales,
                                                        hasSingletonBases=True)

    @staticmethod
    def restore(filePath):
        with open(filePath, 'rb') as f:
            return pickle.load(f)

    def save(self, filePath):
        with open(filePath, 'wb') as f:
            pickle.dump(self, f, protocol=pickle.HIGHEST_PROTOCOL)

    def getNbLevels(self):
        return len(self.scales)

    def getRawDictionary(self, level):
        assert level >= 0 and level < self.getNbLevels()
        return self.dictionaries[level]

    def getBaseDictionary(self):
        return self.dictionaries[0]

    def getMultiscaleDictionaries(self):
        return self.representations
