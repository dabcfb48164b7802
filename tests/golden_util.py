"""Helpers to read the golden fixtures written by tools/make_golden.py (real-reference outputs)."""
import os

import numpy as np
import scipy.sparse

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return _cache[name]


def small_case_names():
    return [str(n) for n in load('cmp_small.npz')['names']]


def small_case(name):
    """Returns (x, D, kwargs, expected) for one case of cmp_small.npz."""
    z = load('cmp_small.npz')
    kw = {}
    for key in ('nbNonzeroCoefs', 'toleranceResidualScale', 'toleranceSnr', 'minCoefficients'):
        full = '%s__%s' % (name, key)
        if full in z:
            v = z[full]
            kw[key] = int(v) if key == 'nbNonzeroCoefs' else float(v)
    if name + '__nbBlocks' in z:
        nb = int(z[name + '__nbBlocks'])
        kw['nbBlocks'] = 'auto' if nb == -1 else nb
    if name + '__weights' in z:
        kw['weights'] = z[name + '__weights']
    exp = dict(t=z[name + '__t'], k=z[name + '__k'], c=z[name + '__c'], residual=z[name + '__residual'],
               row=z[name + '__csc_row'], col=z[name + '__csc_col'], data=z[name + '__csc_data'])
    return z[name + '__x'], z[name + '__D'], kw, exp


def csc_triplets(m):
    m = scipy.sparse.coo_matrix(m)
    order = np.lexsort((m.row, m.col))
    return m.row[order].astype(np.int32), m.col[order].astype(np.int32), m.data[order].astype(np.float64)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
