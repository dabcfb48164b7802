"""Synthetic workloads for the matching-pursuit hot path (BASELINE.json configs).

BASELINE.json gives shapes only; the reference ships no dataset for this path.  The inputs are
therefore generated here, reproducibly, from numpy's frozen legacy generator
(`np.random.RandomState`, stream-compatible across numpy releases) with one independent
stream per signal, so that a rank generating its own shard of the batch produces exactly the
signals a single process would.

  dictionary : D ~ N(0,1) [K,W], rows L2-normalised (as `normalize`, hsc/utils.py:67-74)
  planted    : sum of `nb_atoms` atoms, k ~ U{0..K-1}, centre p ~ U{W/2-1 .. T-W/2-1},
               c = s*U(0.25, 4.0), s = +-1 (amplitude range of hsc/dataset.py:742),
               plus N(0, noise^2) noise
  noise      : x ~ N(0,1) (worst case: no structure, every selection re-ranks the table)
"""
import hashlib

import numpy as np

from .utils import normalize, centered_span

SEED_BASE = 0x48534300


def make_dictionary(K, W, F=1, seed=0, dtype=np.float32):
    rs = np.random.RandomState((SEED_BASE + 7919 * seed) % (2 ** 32))
    shape = (K, W) if F == 1 else (K, W, F)
    D = rs.standard_normal(shape)
    return np.ascontiguousarray(normalize(D, axis=tuple(range(1, D.ndim))).astype(dtype))


def _signal_rs(seed, index):
    return np.random.RandomState((SEED_BASE + 1000003 * seed + index) % (2 ** 32))


def make_signal(D, T, index, kind='planted', nb_atoms=256, noise=0.01, seed=0, dtype=np.float32,
                return_events=False):
    """One signal [T] (D 2-D) or [T,F] (D 3-D); `index` selects the per-signal stream."""
    K, W = D.shape[0], D.shape[1]
    feat = () if D.ndim == 2 else (D.shape[2],)
    rs = _signal_rs(seed, index)
    if kind == 'noise':
        x = rs.standard_normal((T,) + feat)
        events = None
    elif kind == 'planted':
        k = rs.randint(0, K, size=nb_atoms)
        p = rs.randint(W // 2 - 1, T - W // 2, size=nb_atoms)
        c = rs.uniform(0.25, 4.0, size=nb_atoms) * rs.choice([-1.0, 1.0], size=nb_atoms)
        x = noise * rs.standard_normal((T,) + feat)
        D64 = D.astype(np.float64)
        for kk, pp, cc in zip(k, p, c):
            s, e, es, ee = centered_span(T, W, int(pp))
            x[s:e] += cc * D64[kk][es:ee]
        events = (p.astype(np.int32), k.astype(np.int32), c)
    else:
        raise ValueError('unknown signal kind: %s' % kind)
    x = np.ascontiguousarray(x.astype(dtype))
    if return_events:
        return x, events
    return x


def make_batch(D, T, first, count, **kw):
    """Signals first .. first+count-1 stacked [count, T(,F)]; shard-invariant by construction."""
    return np.stack([make_signal(D, T, first + i, **kw) for i in range(count)], axis=0)


def digest(*arrays):
    """sha256 over the raw bytes of the arrays (used by the golden fixtures to detect input drift)."""
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


# ------------------------------------------------------------------------------------------------
# BASELINE config 4: a two-level hierarchy with the dimensions of scripts/learn_mlcsc_dataset.py
# (level 0: K0 atoms x W0 taps; level 1: K1 atoms x W1 taps over the K0 level-0 coefficient streams).
#
# The reference LEARNS that level-1 dictionary (k-means on level-0 coefficients); its generator
# (hsc/dataset.py:515-660) cannot draw one this narrow: it asks the sub-patterns of a composite to
# spread over >= 0.45 x scale, and a 16-row window only offers 15 positions against 0.45 x 79.  The
# hierarchy is therefore composed here, in the container's own terms (`fromDecompositions`,
# hsc/dataset.py:196-306): every level-1 atom is `size` level-0 atoms at offsets inside the W1-row
# window, with weights chosen so that BOTH normalisations of the container hold at once -- the raw
# atom has unit norm in coefficient space and the composed pattern has unit norm in signal space.
# Then a level-1 coefficient c means the same thing to the level coder (c x raw atom) and to the
# reconstruction through the input-level representations (c x composed pattern, :1596-1611), and an
# encode of a signal rendered from the hierarchy reconstructs it (a hierarchy whose two norms
# disagree cannot: the reconstruction is then off by the norm ratio of every level-1 atom used).
# ------------------------------------------------------------------------------------------------
def make_hierarchy_parts(K0=256, W0=64, K1=128, W1=16, size=3, seed=0, dtype=np.float32):
    """(base dictionary [K0,W0], [level-1 decompositions], scales [W0, W0+W1-1]): the arguments of
    MultilevelDictionary.fromDecompositions (hsc/dataset.py:196-306)."""
    assert size == 3 and W1 >= 2 and K0 >= size
    rs = np.random.RandomState((SEED_BASE + 104729 * seed + 4) % (2 ** 32))
    D0 = make_dictionary(K0, W0, seed=seed, dtype=dtype)
    D64 = D0.astype(np.float64)
    scales = [W0, W0 + W1 - 1]
    lead = (W0 - 1) // 2
    decompositions = []
    while len(decompositions) < K1:
        idx = rs.permutation(K0)[:size]
        rows = rs.randint(0, W1, size=size)
        if rows.max() - rows.min() < W1 // 2:
            continue                                          # spread over the window, as the reference asks of its own
        # Gram entries of the shifted sub-atoms inside the composite's window (no clipping: rows + W0 <= scale)
        g = np.zeros((size, size))
        for i in range(size):
            for j in range(i + 1, size):
                d = int(rows[j] - rows[i])
                a, b = D64[idx[i]], D64[idx[j]]
                g[i, j] = float(np.dot(a[d:], b[:W0 - d])) if d >= 0 else float(np.dot(a[:W0 + d], b[-d:]))
        a, b = rs.uniform(0.25, 1.0, size=2) * rs.choice([-1.0, 1.0], size=2)
        den = a * g[0, 2] + b * g[1, 2]
        if abs(den) < 1e-3:
            continue
        c = -a * b * g[0, 1] / den                             # sum_{i<j} w_i w_j g_ij = 0  =>  |composed|^2 = |w|^2
        if not (0.2 <= abs(c) <= 1.5):
            continue
        w = np.array([a, b, c])
        w /= np.sqrt(np.sum(w * w))
        decompositions.append([np.zeros(size, dtype=np.int32), idx.astype(np.int64), (rows + lead).astype(np.int64), w.astype(dtype)])
    return D0, [decompositions], scales


def make_hierarchy(K0=256, W0=64, K1=128, W1=16, size=3, seed=0, dtype=np.float32):
    """MultilevelDictionary (without singleton bases) of the config-4 shape.

    NB the reference's centre conventions (utils.py:84-99 for the windows, dataset.py:159-166 for the
    representations) agree with each other only when centre(W1) == centre(scale1) - centre(scale0), which fails
    exactly for an even scale0 followed by an odd scale1 -- e.g. 64 then 79, the scales that give the 16 taps of
    BASELINE config 4: every level-1 pattern is then reconstructed ONE SAMPLE late and the hierarchy cannot
    reproduce its input whatever the coder does (the reference's own scripts use all-even scales, [32, 64, 96]).
    W1=17 (scales [64, 80]) is the nearest consistent shape; W1=16 is kept for parity tests at the exact dims."""
    from .dataset import MultilevelDictionary
    D0, decompositions, scales = make_hierarchy_parts(K0, W0, K1, W1, size, seed, dtype)
    return MultilevelDictionary.fromDecompositions(D0, decompositions, scales)


def make_hierarchy_signal(mld, T, index, rates=(0.012, 0.006), noise=0.01, seed=0, dtype=np.float32):
    """One signal [T] rendered from the hierarchy: Poisson-many events per level (rate per SAMPLE over all atoms of the
    level), atoms uniform, positions uniform where the pattern fits, amplitudes U(0.25, 4.0) as hsc/dataset.py:742,
    plus N(0, noise^2).  Own per-signal stream (shard-invariant); vectorised counterpart of SignalGenerator."""
    rs = np.random.RandomState((SEED_BASE + 15485863 * seed + 31 * index + 4) % (2 ** 32))
    x = noise * rs.standard_normal(T)
    reps = mld.getMultiscaleDictionaries()
    for level, rep in enumerate(reps):
        n = int(rs.poisson(rates[level] * T))
        scale = rep.shape[1]
        k = rs.randint(0, rep.shape[0], size=n)
        p = rs.randint((scale - 1) // 2, T - scale // 2, size=n)
        c = rs.uniform(0.25, 4.0, size=n)
        pos = (p[:, None] - (scale - 1) // 2 + np.arange(scale)[None, :]).reshape(-1)
        np.add.at(x, pos, (c[:, None] * rep[k].astype(np.float64)).reshape(-1))
    return np.ascontiguousarray(x.astype(dtype))


def make_hierarchy_batch(mld, T, first, count, **kw):
    return np.stack([make_hierarchy_signal(mld, T, first + i, **kw) for i in range(count)], axis=0)


# BASELINE.json configs (shapes) -> keyword sets used by bench.py / tests
CONFIGS = {
    # scripts/demo_csc.py-sized plumbing case
    'config1': dict(B=1, T=4096, K=32, W=32, L0=64),
    # headline: single-level CSC
    'config2': dict(B=1024, T=65536, K=256, W=64, L0=256),
}
