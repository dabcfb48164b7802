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


# BASELINE.json configs (shapes) -> keyword sets used by bench.py / tests
CONFIGS = {
    # scripts/demo_csc.py-sized plumbing case
    'config1': dict(B=1, T=4096, K=32, W=32, L0=64),
    # headline: single-level CSC
    'config2': dict(B=1024, T=65536, K=256, W=64, L0=256),
}
