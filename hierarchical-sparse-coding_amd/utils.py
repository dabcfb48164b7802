"""Host-side index helpers with the reference's names and conventions (hsc/utils.py:67-161).

These are the boundary-clipped, centre-indexed window operations every kernel of the engine
reproduces on the device; the Python versions exist for callers of the reference API
(`reconstructSignal`, LoCOMP-style subclasses, tests).
"""
import numpy as np


def normalize(X, axis=None):
    """L2-normalise along `axis` (default: all but the first); zero rows stay zero (utils.py:67-74)."""
    assert X.ndim >= 1
    if axis is None and X.ndim > 1:
        axis = tuple(range(1, X.ndim))
    norms = np.sqrt(np.sum(np.square(X), axis=axis, keepdims=True))
    norms = np.where(norms > 0.0, norms, np.ones_like(norms))
    return X / norms


def centered_span(length, width, t):
    """Clipped support of a width-`width` element centred at `t` in a signal of `length` samples.

    Even width covers t-(width/2-1) .. t+width/2, odd width t-width//2 .. t+width//2
    (utils.py:84-99).  Returns (start, end, estart, eend): signal[start:end] <-> element[estart:eend];
    end <= start when there is no overlap.
    """
    lo = t - (width - 1) // 2
    hi = t + width // 2 + 1
    start = max(0, lo)
    end = min(length, hi)
    return start, end, start - lo, width - (hi - end)


def peek(signal, width, t):
    """utils.py:76-101"""
    start, end, _, _ = centered_span(signal.shape[0], width, t)
    if end - start > 0:
        return signal[start:end]
    return np.array([], dtype=signal.dtype)


def overlapAdd(signal, element, t, copy=False):
    """utils.py:103-131"""
    if copy:
        signal = np.copy(signal)
    start, end, es, ee = centered_span(signal.shape[0], element.shape[0], t)
    if end - start > 0:
        signal[start:end] += element[es:ee]
    return signal


def overlapReplace(signal, element, t, copy=False):
    """utils.py:133-161"""
    if copy:
        signal = np.copy(signal)
    start, end, es, ee = centered_span(signal.shape[0], element.shape[0], t)
    if end - start > 0:
        signal[start:end] = element[es:ee]
    return signal
