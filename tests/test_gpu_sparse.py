"""GPU parity (-m gpu) of the sparsity-aware multi-feature kernels (levels >= 1 of the hierarchical
encoder, hsc/modeling.py:1427-1492) against the CPU oracle, bit for bit, on each of their three
correlation strategies (and the three ways of finding a window's non-zeros: per-row feature lists,
row-occupancy bitmap + scan, plain scan): sparse window x sparse dictionary pairing (default for level dictionaries),
per-atom dictionary lists (HSCMP_NO_PAIRING), gathered window x dense dictionary (HSCMP_NO_DICT_LISTS)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = {'paired': {}, 'atom_lists': {'HSCMP_NO_PAIRING': '1'},
            'gathered': {'HSCMP_NO_DICT_LISTS': '1', 'HSCMP_FORCE_GATHERED': '1'},
            'dense_dictionary': {'HSCMP_NO_DICT_LISTS': '1'},        # sparse initial correlation + dense LDS-staged loop
            'paired_row_scan': {'HSCMP_NO_ROW_LISTS': '1'},
            'paired_no_rowbits': {'HSCMP_NO_ROW_LISTS': '1', 'HSCMP_NO_ROWBITS': '1'},
            'packed': {'HSCMP_SPARSE_PACKED': '1'}}            # the four-workgroups-per-CU build of the loop (batches > 2 x CUs)


def _level_case(seed, T, F, K, W, dtype, nnz_atom=3, density=0.02, singletons=True):
    """A level->=1 shaped problem: sparse [T, F] input, K composite atoms of nnz_atom events each
    (+ F unit singleton atoms at the centre tap, hsc/dataset.py:826-860)."""
    rs = np.random.RandomState(seed)
    D = np.zeros((K, W, F), dtype=dtype)
    for k in range(K):
        for _ in range(nnz_atom):
            D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
        D[k] /= np.sqrt(np.sum(np.square(D[k])))
    if singletons:
        S = np.zeros((F, W, F), dtype=dtype)
        S[np.arange(F), (W - 1) // 2, np.arange(F)] = 1.0
        D = np.concatenate((S, D), axis=0)
    x = np.zeros((T, F), dtype=dtype)
    n = max(1, int(density * T))
    for _ in range(n):                                       # planted composite events + stray singles
        k = rs.randint(0, D.shape[0]); t = rs.randint(0, T); c = rs.uniform(0.5, 2.0) * rs.choice([-1.0, 1.0])
        s, e = max(0, t - (W - 1) // 2), min(T, t - (W - 1) // 2 + W)
        x[s:e] += (c * D[k][s - (t - (W - 1) // 2):e - (t - (W - 1) // 2)]).astype(dtype)
    return x, D


CASES = [
    # seed, T, F, K, W, dtype, kwargs
    (1, 512, 24, 12, 8, np.float64, dict(nbNonzeroCoefs=60)),
    (2, 512, 24, 12, 8, np.float64, dict(toleranceSnr=25.0, nbBlocks=4)),
    (3, 777, 40, 16, 5, np.float64, dict(toleranceSnr=30.0, nbBlocks='auto')),
    (4, 300, 16, 8, 16, np.float32, dict(nbNonzeroCoefs=40)),
    (5, 300, 16, 8, 16, np.float32, dict(toleranceSnr=20.0, nbBlocks=3)),
    (6, 20, 6, 4, 9, np.float64, dict(nbNonzeroCoefs=10)),            # T < 3W-2: multi-bounce reflection
    (7, 1024, 64, 32, 4, np.float64, dict(toleranceResidualScale=0.05, nbBlocks=8)),
    # dense inputs (seed >= 100): the window / pair lists overflow and the fallback chains run
    (100, 200, 16, 8, 16, np.float64, dict(nbNonzeroCoefs=25)),
    (101, 150, 40, 6, 12, np.float32, dict(toleranceSnr=3.0, nbBlocks=2)),
    # a few dense rows in a sparse input (seed 200..): their feature lists overflow, the rows are read densely
    (200, 400, 24, 10, 8, np.float64, dict(nbNonzeroCoefs=50)),
    (201, 400, 24, 10, 8, np.float32, dict(toleranceSnr=25.0, nbBlocks=4)),
]


@pytest.mark.parametrize('variant', sorted(VARIANTS))
@pytest.mark.parametrize('case', range(len(CASES)))
@pytest.mark.parametrize('weighted', [False, True])
def test_level_shaped_problem_vs_oracle(case, variant, weighted, monkeypatch):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    for key, val in VARIANTS[variant].items():
        monkeypatch.setenv(key, val)
    seed, T, F, K, W, dtype, kw = CASES[case]
    x, D = _level_case(seed, T, F, K, W, dtype)
    if 100 <= seed < 200:
        x = np.random.RandomState(seed).standard_normal((T, F)).astype(dtype)
    elif seed >= 200:
        rs = np.random.RandomState(seed)
        for t in (50, 51, 200, 399):
            x[t] = rs.standard_normal(F).astype(dtype)
    kw = dict(kw)
    if weighted:
        w = np.ones(D.shape[0], dtype=dtype)
        w[:F] = 0.9                                         # singletonWeight, modeling.py:1469-1476
        w[F] = 0.0                                          # a muted atom: its score is 0 whatever its coefficient
        kw['weights'] = w
    cmp = ConvolutionalMatchingPursuit()
    coefficients, residual = cmp.computeCoefficients(x, D, **kw)
    t, k, c = cmp.lastResult.events[0]
    name = cmp.lastResult.variant
    coef, res, info = orc.cmp_encode(x, D, **kw)
    assert len(info['t']) > 0
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']), (variant, name)
    assert np.array_equal(c, info['c'])
    assert np.array_equal(residual, res)
    assert (coefficients != coef).nnz == 0


def test_variants_are_the_ones_dispatched(monkeypatch):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    x, D = _level_case(1, 512, 24, 12, 8, np.float64)
    names = {}
    for variant, env in VARIANTS.items():
        with monkeypatch.context() as m:
            for key, val in env.items():
                m.setenv(key, val)
            cmp = ConvolutionalMatchingPursuit()
            cmp.computeCoefficients(x, D, nbNonzeroCoefs=8)
            names[variant] = cmp.lastResult.variant
    assert names['paired'].startswith('dictlist_init+dictlist_loop')
    assert names['atom_lists'].startswith('dictlist_init+dictlist_loop')
    assert names['gathered'].startswith('sparse_init+gathered_loop')
    assert names['dense_dictionary'].startswith('sparse_init+generic_loop')
    assert names['paired_no_rowbits'].startswith('dictlist_init+dictlist_loop')
    assert names['paired_row_scan'].startswith('dictlist_init+dictlist_loop')
    assert names['packed'].startswith('dictlist_init+dictlist_loop')


def test_dense_level_dictionary_takes_the_dense_loop():
    """A dictionary with more than 32 non-zeros per atom gets no lists: its atoms fill the residual, so the
    loop runs the dense LDS-staged chain (the input itself is still sparse: sparse initial correlation)."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    rs = np.random.RandomState(3)
    x, _ = _level_case(8, 400, 12, 6, 8, np.float64)
    D = rs.standard_normal((10, 8, 12))
    D /= np.sqrt(np.sum(np.square(D), axis=(1, 2), keepdims=True))
    cmp = ConvolutionalMatchingPursuit()
    coefficients, residual = cmp.computeCoefficients(x, D, nbNonzeroCoefs=30)
    assert cmp.lastResult.variant.startswith('sparse_init+generic_loop')
    coef, res, info = orc.cmp_encode(x, D, nbNonzeroCoefs=30)
    t, k, c = cmp.lastResult.events[0]
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c'])
    assert np.array_equal(residual, res)


@pytest.mark.parametrize('variant', ['paired', 'paired_row_scan', 'gathered'])
def test_resumed_launches_keep_the_row_bitmap(variant, monkeypatch):
    """A stopCondition callback splits the loop into one launch per round (hscmp_continue): the row
    flags written back by each launch must carry the spans of the atoms subtracted so far."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    for key, val in VARIANTS[variant].items():
        monkeypatch.setenv(key, val)
    x, D = _level_case(11, 600, 20, 10, 8, np.float64)
    rounds = []

    def stop(sequence, residual, coefficients):
        rounds.append(coefficients.nnz)
        return len(rounds) >= 3

    a = ConvolutionalMatchingPursuit()
    ca, ra = a.computeCoefficients(x, D, nbBlocks=4, stopCondition=stop)
    assert len(rounds) == 3
    n = len(a.lastResult.events[0][0])
    from oracle import hsc_oracle as orc
    coef, res, info = orc.cmp_encode(x, D, nbBlocks=4, maxRounds=3)
    assert n == len(info['t'])
    assert np.array_equal(a.lastResult.events[0][0], info['t']) and np.array_equal(a.lastResult.events[0][1], info['k'])
    assert np.array_equal(a.lastResult.events[0][2], info['c'])
    assert np.array_equal(ra, res)


def test_event_list_growth_on_the_sparse_path():
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    x, D = _level_case(12, 600, 20, 10, 8, np.float64)
    kw = dict(toleranceSnr=30.0, nbBlocks=4)
    a = ConvolutionalMatchingPursuit()
    a.computeCoefficientsBatch(x[np.newaxis], D, maxEvents=3, **kw)
    b = ConvolutionalMatchingPursuit()
    b.computeCoefficientsBatch(x[np.newaxis], D, maxEvents=8192, **kw)
    assert len(b.lastResult.events[0][0]) > 3
    assert all(np.array_equal(u, v) for u, v in zip(a.lastResult.events[0], b.lastResult.events[0]))
    assert np.array_equal(a.lastResult.residuals, b.lastResult.residuals)
