"""GPU parity (-m gpu) of the coefficient-slot lookup (hsc/modeling.py:1106-1114: `coefficients[t, k] += c` on a dict
of keys, with the duplicate / non-zero counts that drive the nbNonzeroCoefs rule).  Short slot lists are searched
behind a Bloom filter; from 1024 slots per signal on the loop keeps an open-addressing table in global memory
(csrc/hscmp_kernels.h: slot_find).  HSCMP_SLOT_HASH_MIN moves that threshold: 0 = table from the first atom,
small values = the switch (table built from the existing slots) happens in the middle of a pursuit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(seed, T, K, W, dtype, F=1):
    rs = np.random.RandomState(seed)
    shape = (K, W) if F == 1 else (K, W, F)
    D = rs.standard_normal(shape).astype(dtype)
    D /= np.sqrt(np.sum(np.square(D.reshape(K, -1)), axis=1)).reshape((K,) + (1,) * (D.ndim - 1))
    x = rs.standard_normal((T,) if F == 1 else (T, F)).astype(dtype)
    return x, D.astype(dtype)


def _check(x, D, kw, expect_dups=False):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    cmp = ConvolutionalMatchingPursuit()
    coefficients, residual = cmp.computeCoefficients(x, D, **kw)
    t, k, c = cmp.lastResult.events[0]
    coef, res, info = orc.cmp_encode(x, D, **kw)
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c'])
    assert np.array_equal(residual, res)
    assert (coefficients != coef).nnz == 0
    if expect_dups:
        pairs = set(zip(info['t'].tolist(), info['k'].tolist()))
        assert len(pairs) < len(info['t'])                   # the case does re-select pairs
    return cmp.lastResult.variant


SHAPES = [
    # seed, T, K, W, F, dtype, kwargs  (short signals, many atoms: every case re-selects some (t,k) pairs)
    (1, 150, 4, 9, 1, np.float32, dict(nbNonzeroCoefs=150)),                       # fused body, one atom per round
    (2, 150, 4, 9, 1, np.float64, dict(nbNonzeroCoefs=150, nbBlocks=6)),           # fused body, blocked rounds
    (3, 200, 6, 16, 1, np.float32, dict(toleranceSnr=25.0, nbBlocks='auto')),
    (4, 100, 5, 7, 3, np.float64, dict(nbNonzeroCoefs=120)),                       # step-by-step body (multi-feature)
    (5, 100, 5, 7, 3, np.float32, dict(nbNonzeroCoefs=120, nbBlocks=5)),
    (6, 160, 4, 12, 2, np.float64, dict(toleranceSnr=8.0, nbBlocks=3)),
]


@pytest.mark.parametrize('hash_min', [0, 1, 7, 40])
@pytest.mark.parametrize('generic', [False, True])
@pytest.mark.parametrize('case', range(len(SHAPES)))
def test_table_lookup_matches_the_slot_scan(case, generic, hash_min, monkeypatch):
    seed, T, K, W, F, dtype, kw = SHAPES[case]
    monkeypatch.setenv('HSCMP_SLOT_HASH_MIN', str(hash_min))
    if generic:
        monkeypatch.setenv('HSCMP_FORCE_GENERIC', '1')
    x, D = _problem(seed, T, K, W, dtype, F)
    _check(x, D, kw, expect_dups=True)


@pytest.mark.parametrize('hash_min', [0, 5])
def test_table_lookup_on_the_sparse_level_path(hash_min, monkeypatch):
    from test_gpu_sparse import _level_case
    monkeypatch.setenv('HSCMP_SLOT_HASH_MIN', str(hash_min))
    for seed, kw in ((11, dict(nbNonzeroCoefs=80)), (12, dict(toleranceSnr=30.0, nbBlocks=4))):
        x, D = _level_case(seed, 512, 24, 12, 8, np.float64)
        name = _check(x, D, kw)
        assert name.startswith('dictlist_init+dictlist_loop')


@pytest.mark.parametrize('blocks', [1, 'auto'])
@pytest.mark.parametrize('generic', [False, True])
def test_long_pursuit_crosses_the_default_threshold(blocks, generic, monkeypatch):
    """More than 1024 slots on one signal with the default threshold: the table is built mid-run, and the nnz
    rule (which counts re-selections apart, :1106-1110) stops both sides at the same event."""
    if generic:
        monkeypatch.setenv('HSCMP_FORCE_GENERIC', '1')
    x, D = _problem(21, 1500, 6, 8, np.float32)
    _check(x, D, dict(nbNonzeroCoefs=1800, nbBlocks=blocks), expect_dups=True)


@pytest.mark.parametrize('hash_min', [0, 30])
def test_table_is_rebuilt_after_the_event_lists_grow(hash_min, monkeypatch):
    """hscmp_grow_events re-allocates the table with the lists; the resumed launch rebuilds it from the slots."""
    from hsc_amd import _native
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    monkeypatch.setenv('HSCMP_SLOT_HASH_MIN', str(hash_min))
    x, D = _problem(31, 800, 6, 8, np.float64)
    kw = dict(nbNonzeroCoefs=300, nbBlocks=4)
    cmp = ConvolutionalMatchingPursuit()
    out = cmp.computeCoefficientsBatch(x[None], D, maxEvents=64, **kw)              # grows 64 -> 256 -> 1024
    coefficients, residual = out.coefficients[0], out.residuals[0]
    t, k, c = out.events[0]
    coef, res, info = orc.cmp_encode(x, D, **kw)
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c'])
    assert np.array_equal(residual, res) and (coefficients != coef).nnz == 0


def test_batch_of_signals_each_with_its_own_table(monkeypatch):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    monkeypatch.setenv('HSCMP_SLOT_HASH_MIN', '3')
    rs = np.random.RandomState(5)
    D = rs.standard_normal((6, 8)).astype(np.float32)
    D /= np.sqrt(np.sum(np.square(D), axis=1, keepdims=True))
    xs = rs.standard_normal((9, 400)).astype(np.float32)
    cmp = ConvolutionalMatchingPursuit()
    kw = dict(nbNonzeroCoefs=90, nbBlocks=3)
    out = cmp.computeCoefficientsBatch(xs, D, **kw)
    coefs, residuals = out.coefficients, out.residuals
    for b in range(xs.shape[0]):
        coef, res, info = orc.cmp_encode(xs[b], D, **kw)
        t, k, c = out.events[b]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c'])
        assert np.array_equal(residuals[b], res) and (coefs[b] != coef).nnz == 0
