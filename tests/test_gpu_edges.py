"""GPU parity (-m gpu) on degenerate inputs, against the CPU oracle: all-zero signals, single samples,
single-tap / single-atom dictionaries, signals shorter than the filters, constant signals (ties everywhere),
multi-feature inputs with a single feature row, batches mixing converged and running signals."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(x, D, **kw):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    cmp = ConvolutionalMatchingPursuit()
    coefficients, residual = cmp.computeCoefficients(x, D, **kw)
    coef, res, info = orc.cmp_encode(x, D, **kw)
    t, k, c = cmp.lastResult.events[0]
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), cmp.lastResult.variant
    assert np.array_equal(residual, res)
    assert residual.shape == res.shape and residual.dtype == res.dtype
    assert (coefficients != coef).nnz == 0
    assert cmp.lastResult.stop_reasons()[0] == info['stop']
    return info


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_zero_signal_stops_before_the_first_selection(dtype):
    rs = np.random.RandomState(0)
    D = rs.standard_normal((8, 16)).astype(dtype)
    info = _check(np.zeros(300, dtype=dtype), D, nbNonzeroCoefs=5)
    assert len(info['t']) == 0
    info = _check(np.zeros((300, 3), dtype=dtype), rs.standard_normal((4, 5, 3)).astype(dtype), toleranceSnr=10.0, nbBlocks=4)
    assert len(info['t']) == 0


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('T,K,W', [(1, 3, 1), (2, 1, 1), (5, 4, 9), (9, 2, 9), (17, 1, 4), (64, 5, 1), (33, 3, 32)])
def test_tiny_and_short_signals(T, K, W, dtype):
    rs = np.random.RandomState(T * 100 + K * 10 + W)
    D = rs.standard_normal((K, W)).astype(dtype)
    D /= np.sqrt(np.sum(np.square(D), axis=1, keepdims=True))
    x = rs.standard_normal(T).astype(dtype)
    _check(x, D, nbNonzeroCoefs=4)
    if T >= 4:                                   # (a block size of 0 is an error in the reference as well)
        _check(x, D, nbNonzeroCoefs=3, nbBlocks=2)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_constant_signal_ties_resolve_like_the_reference(dtype):
    """Every interior position has the same score: the first position (then the first atom) must win."""
    D = np.stack([np.ones(8), np.ones(8), -np.ones(8)]).astype(dtype) / np.sqrt(8)
    x = np.ones(200, dtype=dtype)
    info = _check(x, D, nbNonzeroCoefs=6)
    assert info['k'][0] == 0
    _check(x, D, nbNonzeroCoefs=12, nbBlocks='auto')


def test_batch_with_signals_that_converge_at_different_times():
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    rs = np.random.RandomState(5)
    D = rs.standard_normal((16, 24)).astype(np.float32)
    D /= np.sqrt(np.sum(np.square(D), axis=1, keepdims=True))
    xs = np.zeros((5, 1500), dtype=np.float32)
    xs[1, 100:124] = 2.0 * D[3]                                  # one atom: converges at once
    xs[2] = rs.standard_normal(1500)                             # noise: runs into the L0 limit
    xs[3, 700:724] = D[5]; xs[3, 705:729] -= 0.5 * D[7]
    xs[4] = 1e-30                                                # tiny: below eps right away
    cmp = ConvolutionalMatchingPursuit()
    res = cmp.computeCoefficientsBatch(xs, D, nbNonzeroCoefs=40, toleranceSnr=30.0)
    for b in range(5):
        coef, r, info = orc.cmp_encode(xs[b], D, nbNonzeroCoefs=40, toleranceSnr=30.0)
        t, k, c = res.events[b]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), b
        assert np.array_equal(res.residuals[b], r)
        assert res.stop_reasons()[b] == info['stop']


def test_non_converging_pursuit_raises_instead_of_growing_forever():
    """Two identical feature planes of a constant signal: after the first atom the reference's reflect-padded
    re-correlation (modeling.py:1046) keeps re-selecting position 0 with the same coefficient, nnz never
    grows and the reference loops forever.  The engine enlarges its event lists up to a bound, then raises."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from hsc_amd._native import HscmpError
    D = np.stack([np.ones(8), np.ones(8), -np.ones(8)]) / np.sqrt(8)
    x = np.ones((200, 2))
    with pytest.raises(HscmpError, match='does not converge'):
        ConvolutionalMatchingPursuit().computeCoefficients(x, np.stack([D, D], axis=2), nbNonzeroCoefs=6)


OVERFLOW_CASES = [((400, 1, 8, 16), 1), ((400, 1, 8, 16), 4), ((37, 2, 1, 11), 1), ((37, 2, 1, 11), 'auto'), ((300, 2, 5, 9), 1),
                  ((300, 12, 6, 9, 'sparse'), 4), ((2000, 1, 32, 17), 4)]


@pytest.mark.parametrize('shape,blocks', OVERFLOW_CASES)
@pytest.mark.parametrize('rp', ['1', '0'])
def test_overflowing_residual_ends_without_a_wild_index(shape, blocks, rp, monkeypatch):
    """Float32 signals so large that the first coefficients overflow: correlations become inf / NaN, and every score of a
    row can be NaN.  What the reference does then is not pinned (np.argmax over NaNs), but every loop family must still
    hand valid positions and atoms on: the run ends with a stop reason or with the engine's 'does not converge' error --
    not with a memory fault (which is how the per-row arg-max of the sparse policy used to end, INT_MAX as the atom)."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from hsc_amd._native import HscmpError
    if blocks != 1:
        monkeypatch.setenv('HSCMP_RP', rp)
    elif rp == '0':
        pytest.skip('single arg-max rounds have one loop form')
    rs = np.random.RandomState(5)
    T, F, K, W = shape[:4]
    if len(shape) > 4:
        D = np.zeros((K, W, F), dtype=np.float32)
        for k in range(K):
            for _ in range(3):
                D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5)
    else:
        D = rs.standard_normal((K, W, F)).astype(np.float32)
    D /= np.sqrt(np.sum(np.square(D), axis=(1, 2), keepdims=True))
    x = (rs.standard_normal((T, F)) * 1e38).astype(np.float32)
    if F == 1:
        x, D = x[:, 0], D[:, :, 0]
    cmp = ConvolutionalMatchingPursuit()
    try:
        with np.errstate(all='ignore'):
            cmp.computeCoefficients(x, D, nbNonzeroCoefs=50, nbBlocks=blocks, minCoefficients=None)
    except HscmpError as ex:
        assert 'does not converge' in str(ex)
        return
    t, k, c = cmp.lastResult.events[0]
    assert len(t) > 0 and t.min() >= 0 and t.max() < T and k.min() >= 0 and k.max() < D.shape[0]


@pytest.mark.parametrize('shape,blocks', OVERFLOW_CASES)
def test_overflowing_residual_in_the_locomp_loop(shape, blocks):
    """The same inputs through the LoCOMP loop (dense, sparse and matrix-core policies): NaN Gram entries, pivots and
    scores must still leave valid positions, atoms and slots -- the run ends with a stop reason or 'does not converge'."""
    import logging
    from hsc_amd.modeling import LoCOMP
    from hsc_amd._native import HscmpError
    rs = np.random.RandomState(5)
    T, F, K, W = shape[:4]
    if len(shape) > 4:
        D = np.zeros((K, W, F), dtype=np.float32)
        for k in range(K):
            for _ in range(3):
                D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5)
    else:
        D = rs.standard_normal((K, W, F)).astype(np.float32)
    D /= np.sqrt(np.sum(np.square(D), axis=(1, 2), keepdims=True))
    x = (rs.standard_normal((T, F)) * 1e38).astype(np.float32)
    if F == 1:
        x, D = x[:, 0], D[:, :, 0]
    coder = LoCOMP()
    logging.disable(logging.WARNING)
    try:
        with np.errstate(all='ignore'):
            res = coder.computeCoefficientsBatch(x[np.newaxis], D, nbNonzeroCoefs=50, nbBlocks=blocks, minCoefficients=None)
    except HscmpError as ex:
        assert 'does not converge' in str(ex)
        return
    finally:
        logging.disable(logging.NOTSET)
    t, k, c = res.events[0]
    assert 'locomp' in res.variant and len(t) > 0 and t.min() >= 0 and t.max() < T and k.min() >= 0 and k.max() < D.shape[0]


def test_long_signals():
    """A million samples on the MFMA path (segment maxima at their largest segment size), and a multi-feature input
    longer than the row-bitmap / sparse-initial-correlation limit (262144 rows): generic initial correlation, sparse loop."""
    rs = np.random.RandomState(17)
    T = 1000003
    D = rs.standard_normal((8, 16)).astype(np.float32)
    D /= np.sqrt(np.sum(np.square(D), axis=1, keepdims=True))
    x = (0.01 * rs.standard_normal(T)).astype(np.float32)
    for p in (0, 7, 500000, 999990, T - 1, T - 17, 123456):
        k = rs.randint(0, 8); lo = p - 7; s, e = max(0, lo), min(T, lo + 16)
        x[s:e] += (rs.uniform(1.0, 3.0) * D[k][s - lo:e - lo]).astype(np.float32)
    _check(x, D, nbNonzeroCoefs=30)
    _check(x, D, nbNonzeroCoefs=40, nbBlocks='auto')
    _check(x, D, nbNonzeroCoefs=40, nbBlocks=8)                  # blocks of 125 000 samples: selection through the segment maxima
    _check(x[:300001].astype(np.float64), D.astype(np.float64), nbNonzeroCoefs=25)
    _check(x[:300001].astype(np.float64), D.astype(np.float64), nbNonzeroCoefs=25, nbBlocks=5)
    T2, F = 300000, 4
    D2 = np.zeros((6, 8, F))
    for k in range(6):
        for _ in range(3):
            D2[k, rs.randint(0, 8), rs.randint(0, F)] = rs.uniform(0.5, 1.5)
        D2[k] /= np.sqrt(np.sum(np.square(D2[k])))
    S = np.zeros((F, 8, F)); S[np.arange(F), 3, np.arange(F)] = 1.0
    D2 = np.concatenate((S, D2), axis=0)
    x2 = np.zeros((T2, F))
    for _ in range(400):
        x2[rs.randint(0, T2), rs.randint(0, F)] = rs.uniform(0.5, 2.0)
    x2[0, 1] = 1.5; x2[T2 - 1, 2] = -1.25
    _check(x2, D2, nbNonzeroCoefs=120)


BOUNDARY_SHAPES = [
    # (T, K, W, F): around the dispatch thresholds -- MFMA image size (64 KB f32 / 128 KB f64), W <= 128, T >= 3W-2,
    # 32- / 16-atom groups, segment size steps (T = 512 * 64, 512 * 128, 512 * 256)
    (400, 255, 64, 1), (400, 256, 64, 1), (400, 257, 64, 1), (300, 512, 32, 1), (300, 1024, 16, 1), (700, 300, 64, 1),
    (600, 33, 127, 1), (600, 33, 128, 1), (600, 33, 129, 1), (900, 9, 200, 1),
    (3 * 40 - 3, 12, 40, 1), (3 * 40 - 2, 12, 40, 1), (3 * 40 - 1, 12, 40, 1), (3 * 41 - 3, 12, 41, 1), (3 * 41 - 2, 12, 41, 1),
    (32768, 5, 8, 1), (32769, 5, 8, 1), (65536, 5, 8, 1), (65537, 5, 8, 1), (131072, 5, 8, 1), (131073, 5, 8, 1),
    (500, 17, 9, 2), (500, 16, 9, 33), (260, 40, 5, 64),
]


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('shape', BOUNDARY_SHAPES)
def test_shapes_around_the_dispatch_thresholds(shape, dtype):
    T, K, W, F = shape
    rs = np.random.RandomState(T * 7 + K * 3 + W + F)
    D = rs.standard_normal((K, W) if F == 1 else (K, W, F)).astype(dtype)
    D /= np.sqrt(np.sum(np.square(D), axis=tuple(range(1, D.ndim)), keepdims=True))
    x = (0.05 * rs.standard_normal((T,) if F == 1 else (T, F))).astype(dtype)
    D3 = D.reshape((K, W, -1)); x2 = x.reshape((T, -1))
    for p in (0, T - W, T // 2, T // 3):
        if 0 <= p <= T - W:
            x2[p:p + W] += (rs.uniform(0.8, 2.0) * D3[rs.randint(0, K)]).astype(dtype)
    _check(x, D, nbNonzeroCoefs=14)
    _check(x, D, nbNonzeroCoefs=14, nbBlocks=4)


@pytest.mark.parametrize('xdt,ddt', [(np.float32, np.float64), (np.float64, np.float32), (np.int32, np.float64)])
def test_mixed_input_dtypes(xdt, ddt):
    """The compute dtype is numpy's result type of signal and dictionary, float32 only when both are float32
    (modeling.py:1059-1067 converts nothing: numpy promotes inside the products)."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    rs = np.random.RandomState(3)
    D = rs.standard_normal((6, 12)).astype(ddt)
    D /= np.sqrt(np.sum(np.square(D), axis=1, keepdims=True))
    x = (rs.standard_normal(200) * 4).astype(xdt)
    cmp = ConvolutionalMatchingPursuit()
    coefficients, residual = cmp.computeCoefficients(x, D, nbNonzeroCoefs=10)
    dt = np.float32 if (np.dtype(xdt) == np.float32 and np.dtype(ddt) == np.float32) else np.float64
    coef, res, info = orc.cmp_encode(x.astype(dt), D.astype(dt), nbNonzeroCoefs=10)
    t, k, c = cmp.lastResult.events[0]
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c.astype(np.float64), info['c'].astype(np.float64))
    assert (coefficients != coef).nnz == 0
