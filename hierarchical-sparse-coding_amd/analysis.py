"""Information-rate bookkeeping the signal generator depends on (reference: hsc/analysis.py:37-101).

Only the closed-form rate model is provided (bits per event x Poisson rates, redistributed down the
decomposition tree); the empirical counters, plots and the rest of hsc/analysis.py are outside the
matching-pursuit path.
"""
import collections.abc

import numpy as np


def calculateBitForDatatype(dtype):
    """Bits of one amplitude of `dtype` (hsc/analysis.py:37-44): sign + exponent + fraction, or the
    integer width."""
    dtype = np.dtype(dtype)
    if np.issubdtype(dtype, np.floating):
        info = np.finfo(dtype)
        return 1 + info.iexp + info.nmant
    if np.issubdtype(dtype, np.integer):
        return np.iinfo(dtype).bits
    raise Exception('Unsupported datatype: %s' % (str(dtype)))


def calculateBitForLevels(multilevelDict, sequenceLength, dtype=np.float32):
    """Bits of one event per level: level index + atom index + time index + amplitude
    (hsc/analysis.py:46-57)."""
    atom_bits = np.ceil(np.log(multilevelDict.counts) / np.log(2))
    level_bits = np.ceil(np.log(len(multilevelDict.scales)) / np.log(2))
    time_bits = np.ceil(np.log(sequenceLength) / np.log(2))
    return level_bits + atom_bits + time_bits + calculateBitForDatatype(dtype)


def calculateInformationRate(multilevelDict, rates, sequenceLength, dtype=np.float32):
    """Average bit/sample of independent Poisson event streams (hsc/analysis.py:59-70)."""
    assert len(rates) == multilevelDict.getNbLevels()
    bits = calculateBitForLevels(multilevelDict, sequenceLength, dtype)
    total = 0.0
    for level in range(multilevelDict.getNbLevels()):
        total += np.sum(rates[level] * bits[level])
    return total


def calculateMultilevelInformationRates(multilevelDict, rates, sequenceLength, dtype=np.float32):
    """Bit/sample when the events are expressed at level L, L-1, ..., 0: the rate of every atom of the
    top level is handed down to the atoms of its decomposition, level by level
    (hsc/analysis.py:72-101).  Returns one figure per level, index 0 = everything at the base level.
    (As in the reference, the per-event bit budget is always the float32 one.)"""
    nbLevels = multilevelDict.getNbLevels()
    assert len(rates) == nbLevels
    if not isinstance(rates[0], collections.abc.Iterable):
        rates = [rates[level] * np.ones(multilevelDict.counts[level]) for level in range(nbLevels)]
    out = []
    for level in reversed(range(nbLevels)):
        out.append(calculateInformationRate(multilevelDict, rates, sequenceLength))
        if level > 0:
            for n, (rate, entry) in enumerate(zip(rates[level], multilevelDict.decompositions[level - 1])):
                for l, i in zip(entry[0], entry[1]):
                    rates[l][i] += rate
                rates[level][n] = 0.0
            assert np.allclose(np.sum(rates[level]), 0.0)
    return np.array(out)[::-1]
