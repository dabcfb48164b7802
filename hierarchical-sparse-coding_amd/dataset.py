"""Host-side data model around the hierarchical encoder (reference: hsc/dataset.py).

* the container the encoder reads -- raw dictionaries per level, their input-level representations,
  singleton ("pass-through") bases (hsc/dataset.py:110-410, 826-869) and the event wire format
  (:798-824);
* the synthetic data the reference's experiments run on: Perlin-noise base
  atoms, random decompositions for the higher levels (:412-676) and Poisson event streams rendered to
  signals (:678-796).  Every generator draws from a numpy RandomState in the reference's order, so a
  seeded run reproduces the reference's dictionary and events bit for bit
  (tests/test_dataset_synthesis.py).
"""
import collections.abc
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


class _ReferenceUnpickler(pickle.Unpickler):
    """Unpickler for dictionary files written by the reference (module `hsc.dataset`, Python 2) or by this package."""

    def __init__(self, f):
        pickle.Unpickler.__init__(self, f, encoding='latin1')

    def find_class(self, module, name):
        if module in ('hsc.dataset', 'hsc_amd.dataset') and name == 'MultilevelDictionary':
            return MultilevelDictionary
        if module == 'copy_reg':                      # Python-2 name of copyreg (old-style instance reconstruction)
            module = 'copyreg'
        if module == '__builtin__':
            module = 'builtins'
        return pickle.Unpickler.find_class(self, module, name)


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

    @classmethod
    def fromDecompositions(cls, baseDict, decompositions, scales, hasSingletonBases=False):
        """Raw per-level dictionaries and input-level representations from a base dictionary and, per
        higher level, a list of [levels, indices, positions, coefficients] decompositions
        (hsc/dataset.py:196-306).  Decompositions that reach below the previous level need the
        singleton bases and get them."""
        assert decompositions is not None and len(decompositions) > 0
        assert len(scales) > 0
        nbLevels = len(scales)
        counts = np.array([baseDict.shape[0]] + [len(d) for d in decompositions], dtype=int)
        crossLevel = any(not np.array_equal(entry[0], (level - 1) * np.ones_like(entry[0]))
                         for level in range(1, nbLevels) for entry in decompositions[level - 1])
        lead = [(int(sc) - 1) // 2 for sc in scales]                  # input-level -> level-relative position
        if crossLevel:
            empty = [baseDict] + [np.zeros((counts[level], int(scales[level]), counts[level - 1]), dtype=baseDict.dtype)
                                  for level in range(1, nbLevels)]
            dictionaries = addSingletonBases(empty)
            hasSingletonBases = True
            for level in range(1, nbLevels):
                for entry in decompositions[level - 1]:
                    entry[3] /= np.sqrt(np.sum(np.square(entry[3])))                  # (in place, as the reference)
                    for l, i, t, c in zip(*entry):
                        dictionaries[level][int(np.sum(counts[:l])) + i, t - lead[l]] = c     # :239-253
        else:
            widths = scalesToWindowSizes(scales)
            dictionaries = [baseDict]
            for level in range(1, nbLevels):
                atoms = []
                for _, indices, positions, coefficients in decompositions[level - 1]:
                    atom = np.zeros((widths[level], counts[level - 1]), dtype=coefficients.dtype)
                    atom[positions - lead[level - 1], indices] = coefficients
                    atom /= np.sqrt(np.sum(np.square(atom)))
                    atoms.append(atom)
                dictionaries.append(np.stack(atoms))
        representations = [baseDict]
        for scale, levelDecompositions in zip(scales[1:], decompositions):
            rendered = []
            for levels, indices, positions, coefficients in levelDecompositions:
                signal = np.zeros(int(scale), dtype=baseDict.dtype)
                for l, i, t, c in zip(levels, indices, positions, coefficients):
                    overlapAdd(signal, element=c * representations[l][i, :], t=int(t), copy=False)
                signal /= np.sqrt(np.sum(np.square(signal)))
                rendered.append(signal)
            representations.append(np.stack(rendered))
        return cls(dictionaries, scales, representations, decompositions, hasSingletonBases)

    def upToLevel(self, level):
        """The first level+1 levels as a dictionary of their own (hsc/dataset.py:308-315)."""
        assert level < self.getNbLevels()
        if level == 0:
            return MultilevelDictionary.fromRawDictionaries(self.dictionaries[:1], self.scales[:1], self.hasSingletonBases)
        return MultilevelDictionary.fromDecompositions(self.dictionaries[0], self.decompositions[:level],
                                                       self.scales[:level + 1], self.hasSingletonBases)

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
        """hsc/dataset.py:378-387.  Reads dictionaries saved by this package AND by the reference itself: the reference
        pickles an `hsc.dataset.MultilevelDictionary` (Python 2 cPickle, protocol 2), so the class path is mapped to this
        module and Python-2 byte strings (numpy array buffers) are decoded as latin-1."""
        import os
        filePath = os.path.abspath(filePath)
        _, fmt = os.path.splitext(filePath)
        if fmt != '.pkl' and fmt != '.p':
            raise Exception('Unsupported format: %s' % (fmt))
        with open(filePath, 'rb') as f:
            return _ReferenceUnpickler(f).load()

    def save(self, filePath):
        """hsc/dataset.py:389-396"""
        import os
        filePath = os.path.abspath(filePath)
        _, fmt = os.path.splitext(filePath)
        if fmt != '.pkl' and fmt != '.p':
            raise Exception('Unsupported format: %s' % (fmt))
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


# ------------------------------------------------------------------------------------------------
# synthesis
# ------------------------------------------------------------------------------------------------
_PERLIN_TABLE = (
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21,
    10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149,
    56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122,
    60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132,
    187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250,
    124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183,
    170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110,
    79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81,
    51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4,
    150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180, 151)
# Ken Perlin's reference permutation (public domain table of the classic "improved noise"), 256 entries
# plus the wrap-around copy of the first, as used by hsc/dataset.py:52-64.


def _rng(rng):
    return np.random if rng is None else rng


class Perlin(object):
    """1-D gradient noise over a shuffled permutation table (hsc/dataset.py:47-108): fractal sum of
    `octaves` layers, each 0.4 * lerp(fade(x), g(i) * x, g(i+1) * (x-1)) on the unit cells."""

    PERM = np.array(_PERLIN_TABLE, dtype=int)

    def __init__(self, rng=None):
        self.perm = np.copy(Perlin.PERM)
        self.rng = rng

    def shuffle(self):
        _rng(self.rng).shuffle(self.perm)            # cumulative: every shuffle starts from the previous order

    def sample(self, x, octaves=1, persistence=0.5, lacunarity=2.0, repeat=1024, base=0):
        total = np.zeros_like(x)
        frequency, amplitude, norm = 1.0, 1.0, 0.0
        for _ in range(octaves):
            total += self._layer(x * frequency, int(repeat * frequency), base) * amplitude
            norm += amplitude
            frequency *= lacunarity
            amplitude *= persistence
        return total / norm

    def _layer(self, x, period, base):
        cell = np.floor(x)
        left = np.mod(cell, period).astype(int)
        right = np.mod(left + 1, period)
        frac = x - cell
        fade = frac * frac * frac * (frac * (frac * 6 - 15) + 10)
        a = self._gradient(self.perm[(left & 255) + base], frac)
        b = self._gradient(self.perm[(right & 255) + base], frac - 1)
        return (a + fade * (b - a)) * 0.4

    @staticmethod
    def _gradient(h, x):
        # slopes 1..8, or -1 when bit 3 of the hash is set (the reference's variant of the table, :104-107)
        slope = np.where(h & 8, -1.0, (h & 7) + 1.0)
        return slope * x


class _Rejection(object):
    """Adaptive similarity threshold shared by both pattern samplers (hsc/dataset.py:450-455,
    485-499): a candidate is rejected when its largest |correlation| with the accepted patterns
    exceeds the current threshold; after `patience` consecutive rejections the threshold moves one
    step up a 64-point ladder from 0.05 to 0.95."""

    def __init__(self, patience):
        self.ladder = np.linspace(0.05, 0.95, 64)
        self.step = 0
        self.patience = patience
        self.sampled = self.rejected = self.streak = 0

    def accept(self, accepted, candidate):
        self.sampled += 1
        if len(accepted) >= 1 and np.max(np.abs(np.dot(accepted, candidate))) > self.ladder[self.step]:
            self.rejected += 1
            self.streak += 1
            if self.streak >= self.patience:
                if self.step >= len(self.ladder) - 1:
                    raise Exception("Unable to find the requested number of patterns: maximum correlation is too high")
                self.step += 1
                self.streak = 0
            return False
        self.streak = 0
        return True


class MultilevelDictionaryGenerator(object):
    """hsc/dataset.py:412-676"""

    def __init__(self, rng=None):
        self.rng = rng

    def generate(self, scales, counts, decompositionSize=4, positionSampling='random', weightSampling='random',
                 multilevelDecomposition=True, maxNbPatternsConsecutiveRejected=100, nonNegativity=False):
        assert len(scales) > 0 and len(counts) > 0 and len(scales) == len(counts)
        scales = np.array(scales, dtype=int)
        counts = np.array(counts, dtype=int)
        baseDict = self._generateBaseDictionary(counts[0], scales[0], maxNbPatternsConsecutiveRejected, nonNegativity)
        if len(counts) == 1:
            return MultilevelDictionary.fromBaseDictionary(baseDict)
        decompositions = self._generateHighLevelDecompositions(baseDict, scales, counts, decompositionSize, positionSampling,
                                                               weightSampling, multilevelDecomposition,
                                                               maxNbPatternsConsecutiveRejected)
        return MultilevelDictionary.fromDecompositions(baseDict, decompositions, scales)

    def _generateBaseDictionary(self, nbPatterns, nbPoints, maxNbPatternsConsecutiveRejected=100, nonNegativity=False):
        """Windowed Perlin-noise atoms, unit norm, mutually decorrelated (hsc/dataset.py:437-513)."""
        assert nbPatterns > 0 and nbPoints > 0 and maxNbPatternsConsecutiveRejected > 0
        rng = _rng(self.rng)
        nbPatterns, nbPoints = int(nbPatterns), int(nbPoints)
        perlin = Perlin(self.rng)
        gate = _Rejection(maxNbPatternsConsecutiveRejected)
        patterns = np.zeros((nbPatterns, nbPoints), dtype=np.float32)
        span, maxOctaves = 5.0, 3
        axis = np.arange(2.0 * nbPoints) * span / nbPoints - 0.5 * span      # twice as long: a random window of it
        window = np.hanning(nbPoints)                                        # is used, so zero crossings differ
        found = 0
        while found < nbPatterns:
            start = rng.randint(low=0, high=nbPoints)
            octaves = rng.randint(low=1, high=maxOctaves + 1)
            perlin.shuffle()
            y = perlin.sample(axis[start:start + nbPoints], octaves)
            if nonNegativity:
                y = np.abs(y)
            y *= window
            y /= np.sqrt(np.sum(np.square(y)))
            if gate.accept(patterns[:found, :], y):
                patterns[found, :] = y
                found += 1
        logger.info("Number of patterns found = %d (%d rejected out of %d sampled)" % (found, gate.rejected, gate.sampled))
        return patterns

    def _samplePositions(self, rng, scales, level, selectedLevels, positionSampling):
        half = [int(scales[l]) // 2 for l in selectedLevels]
        even = [int(scales[l]) % 2 == 0 for l in selectedLevels]
        if positionSampling == 'random':
            # anywhere the sub-pattern still fits inside the new pattern (:571-582)
            return np.array([rng.randint(low=h - 1 if ev else h, high=int(scales[level]) - h) for h, ev in zip(half, even)],
                            dtype=int)
        if positionSampling == 'no-overlap':
            # left to right, the slack is shared out at random (:584-606)
            slack = int(scales[level]) - int(np.sum([scales[l] for l in selectedLevels]))
            assert slack >= 0
            positions, nextMin = [], 0.0
            for h, ev in zip(half, even):
                if ev:
                    position = nextMin + rng.randint(low=h - 1, high=h + slack)
                    slack -= (position - nextMin - (h - 1))
                else:
                    position = nextMin + rng.randint(low=h, high=h + slack + 1)
                    slack -= (position - nextMin - h)
                nextMin = position + h
                assert slack >= 0
                positions.append(position)
            return np.array(positions, dtype=int)
        raise Exception('Unsupported position sampling method: %s' % (positionSampling))

    def _generateHighLevelDecompositions(self, baseDict, scales, counts, decompositionSizes, positionSampling='random',
                                         weightSampling='random', multilevelDecomposition=True,
                                         maxNbPatternsConsecutiveRejected=100):
        """Each higher-level pattern = a few lower-level patterns at random offsets and weights, kept if
        it is not too similar to the ones already found (hsc/dataset.py:515-660)."""
        assert maxNbPatternsConsecutiveRejected > 0
        rng = _rng(self.rng)
        representations = [baseDict]
        decompositions = []
        for level in range(1, len(counts)):
            count = int(counts[level])
            size = decompositionSizes[level] if isinstance(decompositionSizes, collections.abc.Iterable) else int(decompositionSizes)
            gate = _Rejection(maxNbPatternsConsecutiveRejected)
            patterns = np.zeros((count, int(scales[level])), dtype=np.float32)
            found, levelDecompositions = 0, []
            while found < count:
                if multilevelDecomposition:
                    drawn = rng.randint(low=0, high=level, size=size)                     # any lower level
                else:
                    drawn = (level - 1) * np.ones((size,), dtype=int)                     # the previous level only
                pickedLevels, pickedIndices = [], []
                for l in range(level):
                    n = len(np.where(drawn == l)[0])
                    if n > 0:
                        if n > counts[l]:
                            raise Exception('Unable to decompose to %d items at sublevels: dictionary size at level %d is too low (%d)' % (n, l, counts[l]))
                        order = rng.permutation(counts[l]).astype(int)                    # no pattern twice
                        pickedLevels.append(l * np.ones((n,), dtype=int))
                        pickedIndices.append(order[:n])
                pickedLevels = np.concatenate(pickedLevels)
                pickedIndices = np.concatenate(pickedIndices)
                while True:
                    positions = self._samplePositions(rng, scales, level, pickedLevels, positionSampling)
                    if float(np.max(positions) - np.min(positions)) >= 0.45 * scales[level] or size == 1:
                        break                                                            # spread over the new pattern
                if weightSampling == 'random':
                    weights = rng.uniform(low=0.25, high=1.0, size=size).astype(np.float32)
                elif weightSampling == 'constant':
                    weights = np.ones(size, dtype=np.float32)
                else:
                    raise Exception('Unsupported weight sampling method: %s' % (weightSampling))
                weights /= np.sqrt(np.sum(np.square(weights)))
                signal = self._composePattern(level, scales, representations, pickedLevels, pickedIndices, positions, weights)
                if gate.accept(patterns[:found, :], signal):
                    patterns[found, :] = signal
                    found += 1
                    levelDecompositions.append([pickedLevels, pickedIndices, positions, weights])
            logger.info("Number of patterns found = %d (%d rejected out of %d sampled)" % (found, gate.rejected, gate.sampled))
            representations.append(np.array([self._composePattern(level, scales, representations, *entry)
                                             for entry in levelDecompositions], dtype=np.float32))
            decompositions.append(levelDecompositions)
        return decompositions

    def _composePattern(self, level, scales, representations, selectedLevels, selectedIndices, positions, coefficients):
        signal = np.zeros(int(scales[level]), dtype=coefficients.dtype)
        for l, i, t, c in zip(selectedLevels, selectedIndices, positions, coefficients):
            overlapAdd(signal, element=c * representations[l][i, :], t=int(t), copy=False)
        signal /= np.sqrt(np.sum(np.square(signal)))
        return signal


class SignalGenerator(object):
    """Independent Poisson event streams, one per atom of every level, rendered through the
    input-level representations (hsc/dataset.py:678-796)."""

    def __init__(self, multilevelDict, rates, rng=None):
        assert len(rates) == multilevelDict.getNbLevels()
        self.multilevelDict = multilevelDict
        self.rates = rates
        self.rng = rng

    def _estimateOptimalRates(self, minimumCompressionRatio, nbSamples):
        """Largest common scaling of the rates whose level-0 bit rate stays under
        minimumCompressionRatio x the raw sample width (hsc/dataset.py:686-706)."""
        from .analysis import calculateBitForDatatype, calculateMultilevelInformationRates
        dtype = self.multilevelDict.dictionaries[0].dtype
        budget = calculateBitForDatatype(dtype) * minimumCompressionRatio
        for factor in np.linspace(1e-6, 1.0, num=1000)[::-1]:
            scaled = np.copy(self.rates) * factor
            if calculateMultilevelInformationRates(self.multilevelDict, scaled, nbSamples, dtype=dtype)[0] <= budget:
                return scaled
        raise Exception("Unable to find the optimal rates: initial rates are too high")

    def _generateSpikegram(self, rate, maxTime=1.0, continuousTime=True):
        """Homogeneous Poisson process on [0, maxTime] by exponential gaps; the draw that crosses maxTime is
        consumed and dropped (hsc/dataset.py:778-796)."""
        rng = _rng(self.rng)
        t, times = 0.0, []
        while t <= maxTime:
            t = t - np.log(rng.uniform()) / rate
            times.append(t)
        times = np.array(times[:-1])
        if not continuousTime:
            times = np.unique(np.floor(times).astype(int))
        return times

    def generateEvents(self, nbSamples=1000, minimumCompressionRatio=None):
        rng = _rng(self.rng)
        rates = self._estimateOptimalRates(minimumCompressionRatio, nbSamples) if minimumCompressionRatio is not None else self.rates
        dtype = self.multilevelDict.dictionaries[0].dtype
        times_, levels_, indices_, values_ = [], [], [], []
        for level, scale in enumerate(self.multilevelDict.scales):
            levelRates = rates[level]
            # events whose pattern would be cut by a border are dropped (:735-741)
            first = np.floor(scale / 2.0 - 1.0) if scale % 2 == 0 else np.floor(scale / 2.0)
            last = nbSamples - np.floor(scale / 2.0)
            for i in range(len(self.multilevelDict.representations[level])):
                rate = levelRates[i] if isinstance(levelRates, collections.abc.Iterable) else float(levelRates)
                times = self._generateSpikegram(rate, maxTime=nbSamples, continuousTime=False)
                times = times[(times >= first) & (times <= last)]
                values = rng.uniform(low=0.25, high=4.0, size=len(times)).astype(dtype)
                times_.append(times); values_.append(values)
                levels_.append(np.full(times.shape, level, dtype=int)); indices_.append(np.full(times.shape, i, dtype=int))
        t = np.concatenate(times_) if times_ else np.zeros((0,), dtype=int)
        order = np.argsort(t, kind='stable')              # by time; ties keep (level, atom) order like sorted()
        events = np.zeros((len(t),), dtype=EVENT_DTYPE)
        if len(t) > 0:
            events['f0'] = t[order]; events['f1'] = np.concatenate(levels_)[order]
            events['f2'] = np.concatenate(indices_)[order]; events['f3'] = np.concatenate(values_)[order]
        if minimumCompressionRatio is not None:
            return events, rates
        return events

    def generateSignalFromEvents(self, events, nbSamples=None):
        scales = self.multilevelDict.scales
        if nbSamples is None:
            ends = [int(t) + int(scales[l]) // 2 for t, l in zip(events['f0'], events['f1'])]
            nbSamples = max(ends) if ends else 0
            logger.info('Number of samples estimated from the events: %d' % (nbSamples))
        dtype = self.multilevelDict.dictionaries[0].dtype
        signal = np.zeros(nbSamples, dtype=dtype)
        representations = self.multilevelDict.representations
        for t, l, i, c in events:
            overlapAdd(signal, element=c * representations[l][i, :], t=int(t), copy=False)
        return signal
