"""Shape sweep of the MFMA (float32, single feature) path against the CPU oracle, bit for bit:
atom counts that are not multiples of the 32-atom MFMA group, odd / short / long filters (compile-time
and run-time chunk counts), signal lengths that are not multiples of the tile or chunk sizes, weights,
every stop rule and selection mode, atoms at both edges (edge-row flags of the score-only state)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [
    # (T, K, W, kwargs)
    (1000, 100, 33, dict(nbNonzeroCoefs=25)),
    (777, 5, 7, dict(nbNonzeroCoefs=30)),
    (3001, 65, 65, dict(nbNonzeroCoefs=20)),
    (2500, 48, 128, dict(nbNonzeroCoefs=12)),
    (4096, 32, 16, dict(toleranceSnr=15.0)),
    (4100, 33, 24, dict(toleranceSnr=12.0, nbBlocks=7)),
    (5000, 40, 40, dict(toleranceSnr=10.0, nbBlocks='auto')),
    (2048, 96, 64, dict(toleranceResidualScale=0.4)),
    (1500, 20, 9, dict(toleranceResidualScale=0.2, nbBlocks=5)),
    (9000, 256, 64, dict(nbNonzeroCoefs=40)),
    (70000, 64, 32, dict(nbNonzeroCoefs=60)),
]


def _inputs(T, K, W, seed, weights, dtype=np.float32):
    import hsc_amd.synth as synth
    D = synth.make_dictionary(K, W, seed=seed, dtype=dtype)
    x = synth.make_signal(D, T, seed, kind='planted', nb_atoms=max(4, T // 150), noise=0.05, seed=seed).astype(np.float64)
    rs = np.random.RandomState(seed)
    # atoms hanging over both edges
    for p, k, c in [(1, 0, 2.5), (W // 3, K - 1, -3.0), (T - 2, K // 2, 2.0), (T - 1 - W // 4, 1 % K, -2.2)]:
        s, e, es, ee = synth.centered_span(T, W, p)
        x[s:e] += c * D[k][es:ee]
    w = None
    if weights:
        w = (0.5 + rs.random_sample(K)).astype(dtype)
    return x.astype(dtype), D, w


@pytest.mark.parametrize('idx', range(len(SHAPES)))
@pytest.mark.parametrize('weights', [False, True])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_mfma_path_vs_oracle(idx, weights, dtype):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    T, K, W, kw = SHAPES[idx]
    if dtype == np.float64 and K * W * 8 > 128 * 1024:
        pytest.skip('dictionary image larger than the float64 MFMA path holds')
    x, D, w = _inputs(T, K, W, 100 + idx, weights, dtype)
    kw = dict(kw)
    if w is not None:
        kw['weights'] = w
    cmp = ConvolutionalMatchingPursuit()
    coefficients, residual = cmp.computeCoefficients(x, D, **kw)
    assert cmp.lastResult.variant.startswith('mfma_init+mfma_loop'), cmp.lastResult.variant
    coef_o, res_o, info = orc.cmp_encode(x, D, **kw)
    t, k, c = cmp.lastResult.events[0]
    assert len(t) == len(info['t']) and len(t) > 0
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k'])
    assert np.array_equal(c, info['c'])
    assert np.array_equal(residual, res_o)
    assert cmp.lastResult.stop_reasons()[0] == info['stop']
    assert (coefficients != coef_o).nnz == 0


def test_large_dictionary_and_short_signal_route_to_generic():
    """Shapes outside the MFMA kernels (dictionary image > 64 KB, T < 3W-2) must still be exact."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    for (T, K, W) in [(600, 300, 64), (100, 8, 40)]:
        x, D, _ = _inputs(T, K, W, 7, False)
        cmp = ConvolutionalMatchingPursuit()
        cmp.computeCoefficients(x, D, nbNonzeroCoefs=10)
        assert cmp.lastResult.variant.startswith('generic')
        _, res_o, info = orc.cmp_encode(x, D, nbNonzeroCoefs=10)
        t, k, c = cmp.lastResult.events[0]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c'])
        assert np.array_equal(cmp.lastResult.residuals[0], res_o)
