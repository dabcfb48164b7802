"""LoCOMP (hsc/modeling.py:1191-1425) against golden vectors of the real reference.

CPU tests replace the three GPU-backed hooks by the CPU oracle (test infrastructure) and check the
host loop: neighbour search, joint least-squares re-fit, stop rules.  -m gpu tests run the product
path (hooks through the C ABI).  LoCOMP re-fits coefficients through a pseudo-inverse, so values are
compared with a tolerance (the reference's own LoCOMP tests use atol=1e-1); the support must match."""
import numpy as np
import pytest

import golden_util as gu


def _case(name):
    z = gu.load('locomp_small.npz')
    kw = {}
    for key in ('nbNonzeroCoefs', 'toleranceSnr', 'minCoefficients'):
        full = '%s__%s' % (name, key)
        if full in z:
            kw[key] = int(z[full]) if key == 'nbNonzeroCoefs' else float(z[full])
    if name + '__nbBlocks' in z:
        nb = int(z[name + '__nbBlocks'])
        kw['nbBlocks'] = 'auto' if nb == -1 else nb
    if name + '__weights' in z:
        kw['weights'] = z[name + '__weights']
    exp = dict(residual=z[name + '__residual'], row=z[name + '__csc_row'], col=z[name + '__csc_col'], data=z[name + '__csc_data'])
    return z[name + '__x'], z[name + '__D'], kw, exp


def _names():
    return [str(n) for n in gu.load('locomp_small.npz')['names']]


def _check(coder, name):
    x, D, kw, exp = _case(name)
    coefficients, residual = coder.computeCoefficients(x, D, **kw)
    row, col, data = gu.csc_triplets(coefficients)
    assert np.array_equal(row, exp['row']) and np.array_equal(col, exp['col']), 'support differs from the reference'
    tol = 2e-4 if np.result_type(x.dtype, D.dtype) == np.float32 else 1e-8
    assert float(np.max(np.abs(data - exp['data']))) <= tol * max(1.0, float(np.max(np.abs(exp['data']))))
    assert residual.shape == exp['residual'].shape and residual.dtype == exp['residual'].dtype
    assert float(np.max(np.abs(residual.astype(np.float64) - exp['residual']))) <= 10 * tol


class _OracleHooks(object):
    """Mixin replacing the GPU-backed hooks by the CPU oracle (and keeping the call on the host loop: the product's
    computeCoefficients runs the device loop of csrc/hscmp_locomp.h, the host loop is what takes over from it)."""

    def computeCoefficients(self, *args, **kw):
        return self._computeCoefficientsHost(*args, **kw)

    def _selectBestAtoms(self, innerProducts, filterWidth, nbBlocks=1, offset=False, nullCoeffThres=0.0, weights=None):
        from oracle import hsc_oracle as orc
        from hsc_amd.modeling import Atom
        t, k, c = orc.select_best_atoms(innerProducts, filterWidth, nbBlocks, offset, nullCoeffThres, weights)
        return [Atom(int(p), int(f), cc, filterWidth) for p, f, cc in zip(t, k, c)]

    def _updateInnerProducts(self, innerProducts, residual, atoms, D):
        from oracle import hsc_oracle as orc
        for a in atoms:
            orc.update_inner_products(innerProducts, residual, D, a.position)
        return innerProducts

    def _initialInnerProducts(self, residual, D, dt):
        from oracle import hsc_oracle as orc
        return np.ascontiguousarray(orc.convolve1d(residual, D, padding='same'), dtype=dt)


@pytest.mark.parametrize('name', _names())
def test_locomp_host_loop_with_oracle_hooks(name):
    import hsc_amd.locomp as locomp

    class OracleLoCOMP(_OracleHooks, locomp.LoCOMP):
        pass

    _check(OracleLoCOMP(), name)


def test_refit_choice_of_the_hierarchical_encoder():
    """LoCOMP() re-fits on the device; the PER-SIGNAL hierarchical entry builds its level coders on the host loop with the
    reference's own pseudo-inverse (a cascade of levels amplifies last-bit differences between solvers)."""
    from hsc_amd.modeling import LoCOMP, HierarchicalConvolutionalMatchingPursuit, ConvolutionalMatchingPursuit
    from hsc_amd import _native
    assert LoCOMP().refit == 'device' and LoCOMP._method == _native.METHOD_LOCOMP
    assert getattr(ConvolutionalMatchingPursuit(), '_method', _native.METHOD_CMP) == _native.METHOD_CMP
    with pytest.raises(AssertionError):
        LoCOMP(refit='somewhere')
    D = np.eye(4, 5, dtype=np.float32)
    coder = HierarchicalConvolutionalMatchingPursuit()._level_coder(D)
    assert isinstance(coder.approximator, LoCOMP) and coder.approximator.refit == 'host'
    assert isinstance(HierarchicalConvolutionalMatchingPursuit(method='cmp')._level_coder(D).approximator, ConvolutionalMatchingPursuit)
    assert _native.STOP_NAMES[_native.STOP_STALLED] == 'stalled' and _native.STOP_NAMES[_native.STOP_GROUP] == 'group'


@pytest.mark.gpu
@pytest.mark.parametrize('name', _names())
def test_locomp_gpu_vs_reference_golden(name):
    """The device loop (neighbourhood, re-fit, group update inside the greedy-loop kernel) against the reference's goldens."""
    from hsc_amd.modeling import LoCOMP
    coder = LoCOMP()
    _check(coder, name)
    assert 'locomp' in coder.lastResult.variant and 'group' not in coder.lastResult.stop_reasons()


@pytest.mark.gpu
@pytest.mark.parametrize('name', _names())
def test_locomp_host_loop_over_the_table_entry_points(name, monkeypatch):
    """The host loop (hscmp_table_open / _select / _update + np.linalg.pinv) that takes over when a neighbourhood exceeds
    the kernel's capacity or a stopCondition is given: the same goldens."""
    from hsc_amd.modeling import LoCOMP
    monkeypatch.setenv('HSCMP_LOCOMP_HOST', '1')
    _check(LoCOMP(), name)


def _planted(T, K, W, F, dtype, seed, natoms, sparse_dictionary=False):
    rs = np.random.RandomState(seed)
    if sparse_dictionary:
        D = np.zeros((K, W, F))
        for k in range(K):
            for _ in range(3):
                D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
    else:
        D = rs.standard_normal((K, W, F))
    D /= np.sqrt(np.sum(np.square(D), axis=(1, 2), keepdims=True))
    x = 0.01 * rs.standard_normal((T, F)) * (rs.rand(T, F) < (0.05 if sparse_dictionary else 1.0))
    for _ in range(natoms):
        k, t = rs.randint(0, K), rs.randint(0, T - W)
        x[t:t + W] += rs.uniform(0.5, 2.0) * rs.choice([-1.0, 1.0]) * D[k]
    D, x = D.astype(dtype), x.astype(dtype)
    return (x[:, 0], D[:, :, 0]) if F == 1 else (x, D)


BATCH_CASES = [
    # T, K, W, F, dtype, planted atoms, keyword arguments, sparse dictionary
    (700, 12, 16, 1, np.float32, 30, dict(toleranceSnr=20.0), False),
    (700, 12, 16, 1, np.float64, 30, dict(toleranceSnr=25.0, nbBlocks=4), False),
    (900, 20, 9, 1, np.float32, 60, dict(nbNonzeroCoefs=40, nbBlocks='auto'), False),
    (400, 6, 8, 3, np.float64, 25, dict(toleranceSnr=20.0, nbBlocks=3), False),
    (600, 10, 9, 12, np.float64, 40, dict(toleranceSnr=25.0, nbBlocks=4), True),
    (600, 10, 9, 12, np.float64, 40, dict(nbNonzeroCoefs=30), True),
    (500, 8, 17, 1, np.float32, 25, dict(toleranceSnr=15.0, nbBlocks=2, weights='w'), False),
    (2000, 70, 64, 1, np.float32, 40, dict(toleranceSnr=20.0, nbBlocks=5, weights='w'), False),
    (1500, 33, 32, 1, np.float32, 40, dict(toleranceSnr=20.0), False),
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', range(len(BATCH_CASES)))
def test_locomp_device_loop_vs_host_loop(case, monkeypatch):
    """A batch of related signals through the device loop (csrc/hscmp_locomp.h: one workgroup per signal, neighbourhood + float64
    normal equations + group update in the kernel) against the host loop over the table entry points (np.linalg.pinv as in the
    reference): same support, same stop, coefficients and residual within the tolerance of the two solvers.  Dense and sparse
    (LocompSparse) policies, single arg-max and blocked rounds, weights."""
    from hsc_amd.modeling import LoCOMP
    T, K, W, F, dtype, natoms, kw, sp = BATCH_CASES[case]
    x, D = _planted(T, K, W, F, dtype, 100 + case, natoms, sp)
    kw = dict(kw)
    if kw.get('weights') == 'w':
        kw['weights'] = np.linspace(0.6, 1.0, K).astype(dtype)
    rs = np.random.RandomState(case)
    xs = np.stack([x, (0.5 * x).astype(dtype), x[::-1].copy(), (x + 0.02 * rs.standard_normal(x.shape)).astype(dtype), np.zeros_like(x)])
    dev = LoCOMP()
    res = dev.computeCoefficientsBatch(xs, D, **kw)
    assert 'locomp' in res.variant and (not sp or 'dictlist' in res.variant)
    reasons = res.stop_reasons()
    assert reasons.count('group') == 0 and reasons.count('host') <= 1     # (a member that outgrew the signal's group scratch would come back as 'host')
    again = dev.computeCoefficientsBatch(xs, D, **kw)
    if 'mfma' in res.variant:
        # single-feature float32: the re-correlations ran on the matrix cores; the dense form gives the same bits
        assert F == 1 and dtype == np.float32
        monkeypatch.setenv('HSCMP_LOCOMP_NO_MFMA', '1')
        dense = LoCOMP().computeCoefficientsBatch(xs, D, **kw)
        monkeypatch.delenv('HSCMP_LOCOMP_NO_MFMA')
        assert 'mfma' not in dense.variant and np.array_equal(dense.stats, res.stats)
        # ... and so do two signals per workgroup around one dictionary image (what a batch larger than the chip runs)
        for b in range(xs.shape[0]):
            assert (res.coefficients[b] != dense.coefficients[b]).nnz == 0 and np.array_equal(res.residuals[b], dense.residuals[b])
        for pack in ('2', '4'):
            monkeypatch.setenv('HSCMP_LOCOMP_PACK', pack)
            packed = LoCOMP().computeCoefficientsBatch(xs, D, **kw)
            monkeypatch.delenv('HSCMP_LOCOMP_PACK')
            if pack == '4' and ('group' in packed.stop_reasons() or 'host' in packed.stop_reasons()):
                continue                                       # (four per workgroup re-fit groups of at most 32 atoms)
            assert np.array_equal(packed.stats, res.stats), pack
            for b in range(xs.shape[0]):
                assert (res.coefficients[b] != packed.coefficients[b]).nnz == 0 and np.array_equal(res.residuals[b], packed.residuals[b]), (pack, b)
    else:
        assert not (F == 1 and dtype == np.float32 and (W + 7) // 8 in (2, 4, 8))
    monkeypatch.setenv('HSCMP_LOCOMP_HOST', '1')
    tol = 2e-5 if dtype == np.float32 else 1e-9
    for b in range(xs.shape[0]):
        assert (res.coefficients[b] != again.coefficients[b]).nnz == 0 and np.array_equal(res.residuals[b], again.residuals[b])
        ch, rh = LoCOMP().computeCoefficients(xs[b], D, **kw)
        if reasons[b] == 'host':
            assert (res.coefficients[b] != ch).nnz == 0 and np.array_equal(res.residuals[b], rh)
            continue
        a, h = res.coefficients[b].tocsc(), ch.tocsc()
        assert a.shape == h.shape and a.nnz == h.nnz and np.array_equal(a.indices, h.indices) and np.array_equal(a.indptr, h.indptr), (case, b)
        scale = max(1.0, float(np.max(np.abs(h.data))) if h.nnz else 1.0)
        assert (float(np.max(np.abs(a.data - h.data))) if h.nnz else 0.0) <= tol * scale, (case, b)
        assert float(np.max(np.abs(res.residuals[b].astype(np.float64) - rh))) <= 10 * tol * scale, (case, b)


# every seed of tests/test_gpu_fuzz.py::_draw in the range: the draws whose groups are rank deficient (tiny signals under large
# dictionaries: 29, 53, 117 -- reference goldens of exactly those inputs are in tests/test_locomp_hier.py) go through the kernel's
# minimum-norm completion, which is what np.linalg.pinv returns on the host side of this comparison; seed 9 (2 samples, 5 taps) is a
# pursuit that diverges to inf in the reference itself and leaves through the isfinite test below
SOAK_SEEDS = list(range(0, 130))


@pytest.mark.gpu
@pytest.mark.parametrize('i', SOAK_SEEDS)
def test_locomp_random_configuration_device_loop_vs_host_loop(i, monkeypatch):
    """Random shapes, dtypes, stop rules and selection modes (the draws of the greedy loop's fuzz sweep) through the LoCOMP device
    loop and through the host loop: same support, coefficients and residual within the solvers' tolerance."""
    import logging
    import test_gpu_fuzz as fz
    from hsc_amd.modeling import LoCOMP
    x, D, kw = fz._draw(i)
    kw = dict(kw)
    if kw.get('nbNonzeroCoefs', 0) > 40:
        kw['nbNonzeroCoefs'] = 40
    if x.shape[0] > 1500:
        pytest.skip('long signal: the host loop takes seconds')
    logging.disable(logging.WARNING)
    try:
        dev = LoCOMP()
        cd, rd = dev.computeCoefficients(x, D, **kw)
        assert 'locomp' in dev.lastResult.variant
        monkeypatch.setenv('HSCMP_LOCOMP_HOST', '1')
        ch, rh = LoCOMP().computeCoefficients(x, D, **kw)
    finally:
        logging.disable(logging.NOTSET)
    if not np.all(np.isfinite(rh)):
        return                                                   # (a diverging pursuit: inf / NaN comparisons are not pinned)
    a, h = cd.tocsc(), ch.tocsc()
    tol = 1e-4 if np.result_type(x.dtype, D.dtype) == np.float32 else 1e-8
    scale = max(1.0, float(abs(h).max()) if h.nnz else 1.0)
    # (an entry that cancels to exactly 0.0 in one of the two solvers leaves its support: compare values, not patterns)
    assert (float(abs(a - h).max()) if (a.nnz or h.nnz) else 0.0) <= tol * scale, (i, dev.lastResult.variant)
    assert float(np.max(np.abs(rd.astype(np.float64) - rh.astype(np.float64)))) <= 10 * tol * scale, (i, dev.lastResult.variant)


@pytest.mark.gpu
def test_locomp_groups_beyond_the_lds_copy_with_four_signals_per_workgroup(monkeypatch):
    """Four signals per workgroup keep the Gram matrix of a group in LDS up to 32 atoms; a larger group (up to the kernel's 64)
    solves through the signal's global scratch (up to the kernel's 128).  Dense codes -- neighbourhoods of 33+ atoms -- through one, two and four signals
    per workgroup: bit-identical."""
    from hsc_amd.modeling import LoCOMP
    rs = np.random.RandomState(11)
    T, K, W = 1200, 70, 64
    D = rs.standard_normal((K, W)).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=1, keepdims=True))
    xs = rs.standard_normal((6, T)).astype(np.float32)
    kw = dict(toleranceSnr=17.0, nbBlocks=3)
    out = {}
    for pack in ('1', '2', '4'):
        monkeypatch.setenv('HSCMP_LOCOMP_PACK', pack)
        out[pack] = LoCOMP().computeCoefficientsBatch(xs, D, **kw)
        assert 'locomp_mfma' in out[pack].variant and 'group' not in out[pack].stop_reasons(), pack
    monkeypatch.delenv('HSCMP_LOCOMP_PACK')
    # the codes are dense enough: some atom has more than 32 others within the neighbourhood's reach
    c = out['1'].coefficients[0].tocoo()
    t = np.sort(c.row)
    assert (np.searchsorted(t, t + W - 2, side='right') - np.searchsorted(t, t - (W - 2))).max() > 36      # (reach of :1228-1236: about +-W)
    for pack in ('2', '4'):
        assert np.array_equal(out[pack].stats, out['1'].stats), pack
        for b in range(xs.shape[0]):
            assert (out[pack].coefficients[b] != out['1'].coefficients[b]).nnz == 0 and np.array_equal(out[pack].residuals[b], out['1'].residuals[b]), (pack, b)


@pytest.mark.gpu
def test_locomp_neighbourhood_beyond_the_kernel_capacity_goes_to_the_host_loop(monkeypatch):
    """More than 127 previously selected atoms around a new one (short filters, many atoms per position, a demanding SNR).  With the
    group capacity held at the size of the LDS lists (HSCMP_LOCOMP_GROUP_CAP=128; the default, 512, keeps larger groups in the signal's
    global scratch -- next test) the kernel stops the signal with reason 'group' before applying anything of that atom, and the batch
    entry repeats the signal on the host loop -- the result is the host loop's, the other signals keep the device loop's."""
    from hsc_amd.modeling import LoCOMP
    monkeypatch.setenv('HSCMP_LOCOMP_GROUP_CAP', '128')
    rs = np.random.RandomState(4)
    K, W, T, F = 60, 5, 60, 24                                # (a neighbourhood spans ~3W x F = 360 dimensions: room for > 127 atoms)
    D = rs.standard_normal((K, W, F)).astype(np.float64)
    D /= np.sqrt(np.sum(D ** 2, axis=(1, 2), keepdims=True))
    dense = rs.standard_normal((T, F))
    sparse = np.zeros((T, F)); sparse[20:25] = D[3]; sparse[40:45] = -2.0 * D[7]
    xs = np.stack([dense, sparse])
    coder = LoCOMP()
    kw = dict(toleranceSnr=50.0, nbNonzeroCoefs=3000)
    res = coder.computeCoefficientsBatch(xs, D, **kw)
    assert res.stop_reasons()[0] == 'host' and res.stop_reasons()[1] not in ('group', 'host')      # ('host': given up by the kernel, finished by the host loop)
    host = LoCOMP(refit='host')
    ch, rh = host.computeCoefficients(dense, D, **kw)
    assert (res.coefficients[0] != ch).nnz == 0 and np.array_equal(res.residuals[0], rh)
    # the counters of that signal describe the host run
    from hsc_amd import _native
    assert res.stats[0, _native.STAT_NNZ] == ch.nnz and abs(res.energies[0, 1] - float(np.sum(np.square(rh)))) <= 1e-12 * res.energies[0, 0]
    cs, rsd = host.computeCoefficients(sparse, D, **kw)
    assert np.array_equal(res.coefficients[1].tocsc().indices, cs.tocsc().indices)
    assert float(np.max(np.abs(res.coefficients[1].tocsc().data - cs.tocsc().data))) <= 1e-9


@pytest.mark.gpu
def test_locomp_groups_beyond_the_lds_lists_stay_on_the_device():
    """The same dense signal with the default capacity: its neighbourhoods of more than 127 atoms keep their lists, their Gram matrix
    and the pivoted factorisation in the signal's global scratch (all threads of the workgroup), nothing goes to the host loop, and
    the result agrees with the host loop's np.linalg.pinv re-fits."""
    from hsc_amd.modeling import LoCOMP
    rs = np.random.RandomState(4)
    K, W, T, F = 60, 5, 60, 24
    D = rs.standard_normal((K, W, F)).astype(np.float64)
    D /= np.sqrt(np.sum(D ** 2, axis=(1, 2), keepdims=True))
    dense = rs.standard_normal((T, F))
    kw = dict(toleranceSnr=50.0, nbNonzeroCoefs=3000)
    res = LoCOMP().computeCoefficientsBatch(np.stack([dense, dense]), D, **kw)
    assert 'group' not in res.stop_reasons()
    ch, rh = LoCOMP(refit='host').computeCoefficients(dense, D, **kw)
    c = res.coefficients[0].tocoo()
    t = np.sort(c.row)
    assert (np.searchsorted(t, t + W, side='right') - np.searchsorted(t, t - W)).max() > 140       # (groups beyond the LDS lists did occur)
    a, h = res.coefficients[0].tocsc(), ch.tocsc()
    assert a.nnz == h.nnz and np.array_equal(a.indices, h.indices) and np.array_equal(a.indptr, h.indptr)
    assert float(np.max(np.abs(a.data - h.data))) <= 1e-7 * max(1.0, float(np.max(np.abs(h.data))))
    assert (res.coefficients[1] != res.coefficients[0]).nnz == 0 and np.array_equal(res.residuals[0], res.residuals[1])


@pytest.mark.gpu
def test_hierarchical_default_method_is_locomp():
    """HierarchicalConvolutionalMatchingPursuit() defaults to method='locomp' (modeling.py:1429):
    3-level encode against the reference's golden output (case 'd')."""
    from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit, HierarchicalConvolutionalSparseCoder
    from test_hierarchical import _mld, _golden
    z = _golden()
    hcmp = HierarchicalConvolutionalMatchingPursuit()
    assert hcmp.method == 'locomp'
    hcsc = HierarchicalConvolutionalSparseCoder(_mld(), hcmp)
    coefficients, residual = hcsc.encode(z['x'], toleranceSnr=[10.0, 20.0, 20.0], nbBlocks=4, singletonWeight=0.5)
    assert len(coefficients) == 3
    for l, c in enumerate(coefficients):
        row, col, data = gu.csc_triplets(c)
        assert np.array_equal(row, z['case_d__level%d_row' % l]) and np.array_equal(col, z['case_d__level%d_col' % l])
        assert float(np.max(np.abs(data - z['case_d__level%d_data' % l]))) <= 2e-4
    assert float(np.max(np.abs(residual - z['case_d__residual']))) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('F,density', [(7, 0.05), (40, 0.01), (3, 1.0), (300, 0.9)])
def test_multi_feature_table_rows_from_the_non_zero_cells(F, density):
    """Multi-feature tables (hierarchical levels >= 1) are built row by row from the non-zero cells of each row's window
    (table_rows_sparse_kernel): open and in-place updates against the oracle's dense operations, bit for bit -- sparse
    inputs, a dense one, and one whose windows overflow the row list (F=300, 90 % non-zero: the dense chain per row)."""
    from oracle import hsc_oracle as orc
    from hsc_amd import _native
    from hsc_amd.modeling import ConvolutionalMatchingPursuit, Atom
    rs = np.random.RandomState(F)
    T, K, W = 300, 12, 9
    D = rs.standard_normal((K, W, F)) * (rs.rand(K, W, F) < 0.3)
    D /= np.sqrt(np.sum(D ** 2, axis=(1, 2), keepdims=True))
    x = rs.standard_normal((T, F)) * (rs.rand(T, F) < density)
    eng = _native.default_engine(0)
    eng.set_dictionary(D)
    tab = eng.table_open(x)
    ip = np.ascontiguousarray(orc.convolve1d(x, D, padding='same'), dtype=np.float64)
    assert np.array_equal(tab.read(), ip)
    cmp = ConvolutionalMatchingPursuit()
    residual = x.copy()
    for rnd in range(3):
        atoms = [Atom(0, 1, 0.25, W), Atom(T - 1, 2, -0.5, W), Atom(W, 3, 0.125, W), Atom(int(rs.randint(W, T - W)), int(rs.randint(0, K)), 0.7, W),
                 Atom(T - 1 - W, 0, 0.3, W)]
        residual, _ = cmp._updateResidual(residual, 0.0, atoms, D)
        cmp._updateInnerProducts(tab, residual, atoms, D)
        for a in atoms:
            orc.update_inner_products(ip, residual, D, a.position)
        tb, rb = tab.read(), eng.table_read(table=False, residual=True)[1]
        assert np.array_equal(tb, ip), rnd
        assert np.array_equal(rb, residual), rnd


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_device_resident_table_equals_host_table_ops(dtype):
    """hscmp_table_open / _select / _update (the table and the residual stay on the device) against the oracle's
    operations on a host table, bit for bit: initial table, four rounds of select + residual edit + in-place update
    (atoms at the signal ends and inside), blocked and single selection, with and without weights."""
    from oracle import hsc_oracle as orc
    from hsc_amd import _native
    from hsc_amd.modeling import ConvolutionalMatchingPursuit, Atom
    rs = np.random.RandomState(11)
    T, K, W = 700, 24, 17
    D = rs.standard_normal((K, W, 1)).astype(dtype)
    D /= np.sqrt(np.sum(D ** 2, axis=(1, 2), keepdims=True))
    x = rs.standard_normal((T, 1)).astype(dtype)
    eng = _native.default_engine(0)
    eng.set_dictionary(D)
    tab = eng.table_open(x)
    ip = np.ascontiguousarray(orc.convolve1d(x, D, padding='same'), dtype=dtype)
    assert np.array_equal(tab.read(), ip)
    cmp = ConvolutionalMatchingPursuit()
    residual = x.copy()
    weights = (0.5 + rs.rand(K)).astype(dtype)
    for rnd, (nb, w, offs) in enumerate([(1, None, False), (5, None, True), ('auto', weights, False), (3, weights, True)]):
        got = cmp._selectBestAtoms(tab, W, nbBlocks=nb, offset=offs, nullCoeffThres=1e-6, weights=w)
        t, k, c = orc.select_best_atoms(ip, W, nb, offs, 1e-6, w)
        assert [a.position for a in got] == [int(v) for v in t] and [a.index for a in got] == [int(v) for v in k]
        assert np.array_equal(np.array([a.coefficient for a in got], dtype=dtype), c)
        atoms = got[:3] + [Atom(0, 1, 0.25, W), Atom(T - 1, 2, -0.5, W), Atom(W, 3, 0.125, W)]
        residual, _ = cmp._updateResidual(residual, 0.0, atoms, D)
        cmp._updateInnerProducts(tab, residual, atoms, D)
        for a in atoms:
            orc.update_inner_products(ip, residual, D, a.position)
        tab.flush()                                             # (updates are deferred to the next selection: send them now)
        tb, rb = eng.table_read(table=True, residual=True)
        assert np.array_equal(tb, ip), rnd
        assert np.array_equal(rb, residual), rnd


@pytest.mark.gpu
def test_table_calls_without_open_table_fail_loudly():
    from hsc_amd import _native
    eng = _native.Engine(0)
    D = np.eye(4, 5, dtype=np.float32)[:, :, None]
    eng.set_dictionary(D)
    eng._table_T = 16
    with pytest.raises(_native.HscmpError):
        eng.table_select()
    tab = eng.table_open(np.zeros((16, 1), dtype=np.float32))
    with pytest.raises(_native.HscmpError):
        eng.table_update(np.zeros((4, 1), dtype=np.float32), 14, [3])       # samples past the end
    with pytest.raises(_native.HscmpError):
        eng.table_update(np.zeros((4, 1), dtype=np.float32), 0, [16])       # centre outside the signal
    tab2 = eng.table_open(np.ones((16, 1), dtype=np.float32))               # the engine holds one table: the old handle is retired
    with pytest.raises(_native.HscmpError):
        tab.flush()
    tab = tab2
    eng.set_dictionary(2 * D)                                               # a new dictionary drops the table
    with pytest.raises(_native.HscmpError):
        tab.read()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('case', range(len(BATCH_CASES)))
def test_locomp_selections_computed_side_by_side_equal_the_sequential_loop(case, monkeypatch):
    """Blocked rounds whose selections lie more than 5W + 8 samples apart get neighbourhood, normal equations, re-fit (and, with a dense
    dictionary, the subtractions on a private copy of the span) of up to four selections computed at once, one wave each, before they
    are applied in order (locomp_precompute); the rows of such a batch are re-correlated behind its last selection, one wave per
    selection (locomp_rows_deferred), and on sparse dictionaries the rows of a group that does not wait go one wave per quarter
    (RpSparse::rows_listed).  HSCMP_LOCOMP_AHEAD=0 applies the same rounds selection by selection, workgroup-wide: every signal of
    every policy bit for bit -- long signals with few blocks so that the rounds really are spaced."""
    from hsc_amd.modeling import LoCOMP
    T, K, W, F, dtype, natoms, kw, sp = BATCH_CASES[case]
    T = 40 * W * 6
    x, D = _planted(T, K, W, F, dtype, 300 + case, 12 * natoms, sp)
    kw = dict(kw)
    kw['nbBlocks'] = 6
    if kw.get('weights') == 'w':
        kw['weights'] = np.linspace(0.6, 1.0, K).astype(dtype)
    if 'nbNonzeroCoefs' in kw:
        kw['nbNonzeroCoefs'] = 8 * kw['nbNonzeroCoefs']
    rs = np.random.RandomState(case)
    xs = np.stack([x, (0.5 * x).astype(dtype), x[::-1].copy(), (x + 0.02 * rs.standard_normal(x.shape)).astype(dtype)])
    ahead = LoCOMP().computeCoefficientsBatch(xs, D, **kw)           # (default: everything on -- HSCMP_LOCOMP_AHEAD=7)
    monkeypatch.setenv('HSCMP_LOCOMP_AHEAD', '3')                    # side by side, every group's rows at once (a quarter per wave on sparse dictionaries)
    undeferred = LoCOMP().computeCoefficientsBatch(xs, D, **kw)
    monkeypatch.setenv('HSCMP_LOCOMP_AHEAD', '1')                    # side by side, but a group's rows by the whole workgroup
    only_ahead = LoCOMP().computeCoefficientsBatch(xs, D, **kw)
    monkeypatch.setenv('HSCMP_LOCOMP_AHEAD', '0')
    seq = LoCOMP().computeCoefficientsBatch(xs, D, **kw)
    assert np.array_equal(only_ahead.stats, seq.stats) and np.array_equal(only_ahead.residuals, seq.residuals)
    assert np.array_equal(undeferred.stats, seq.stats) and np.array_equal(undeferred.residuals, seq.residuals)
    assert np.array_equal(ahead.stats, seq.stats) and np.array_equal(ahead.energies, seq.energies)
    assert int(ahead.stats[:, 4].sum()) > 4 * xs.shape[0]          # (several selections per signal)
    for b in range(xs.shape[0]):
        assert (ahead.coefficients[b] != seq.coefficients[b]).nnz == 0 and np.array_equal(ahead.residuals[b], seq.residuals[b]), b
        assert all(np.array_equal(u, v) for u, v in zip(ahead.events[b], seq.events[b])), b


@pytest.mark.gpu
def test_locomp_large_groups_down_to_round_off():
    """30 samples x 12 features under 48 atoms of 5 taps, pursued far down: groups of more than 135 atoms (lists, Gram matrix and the
    pivoted factorisation in the signal's global scratch, all threads of the workgroup) that come close to the number of cells they
    span.  At 100 dB (residual 3e-8) device loop and host loop (np.linalg.pinv in float64) agree to 1e-12; at 150 dB the pursuit is in
    round-off (residual 1e-13, the last atoms have coefficients of 1e-7): same atoms of size, coefficients to 1e-5."""
    from hsc_amd.modeling import LoCOMP
    rs = np.random.RandomState(8)
    K, W, T, F = 48, 5, 30, 12
    D = rs.standard_normal((K, W, F)).astype(np.float64)
    D /= np.sqrt(np.sum(D ** 2, axis=(1, 2), keepdims=True))
    x = rs.standard_normal((T, F))
    for snr, tol, floor in ((100.0, 1e-12, 0.0), (150.0, 1e-5, 1e-4)):
        kw = dict(nbNonzeroCoefs=1000, toleranceSnr=snr)
        res = LoCOMP().computeCoefficientsBatch(np.stack([x, x]), D, **kw)
        assert 'group' not in res.stop_reasons() and 'host' not in res.stop_reasons()
        t = np.sort(res.coefficients[0].tocoo().row)
        assert (np.searchsorted(t, t + W, side='right') - np.searchsorted(t, t - W)).max() > 135      # (groups beyond the LDS lists)
        ch, rh = LoCOMP(refit='host').computeCoefficients(x, D, **kw)
        a, h = res.coefficients[0].tocsc(), ch.tocsc()
        scale = float(abs(h).max())
        assert ((abs(a) > floor * scale) != (abs(h) > floor * scale)).nnz == 0, snr
        assert float(abs(a - h).max()) <= tol * scale, snr
        assert (res.coefficients[1] != res.coefficients[0]).nnz == 0


@pytest.mark.gpu
def test_host_loop_sends_edge_updates_at_once():
    """A one-atom dictionary of 14 taps on a 22-sample signal, eight blocks (draw 1353 of the fuzz sweep, found by tools/locomp_soak.py):
    every re-correlation window crosses both signal ends, so the rows read reflected samples (modeling.py:1046) that LATER atoms of the
    round change without re-correlating them -- the reference's table keeps the values computed at the time.  The host loop over the
    device-resident table used to defer all updates of a round to its end (exact only for rows that read the residual as it is) and was
    1e-2 off; an update whose window crosses an end now goes to the device at once.  With one atom in the dictionary LoCOMP has no
    groups (:1241 drops entries of the same index), so the CPU oracle's greedy coder is the reference answer."""
    import test_gpu_fuzz as fz
    from oracle import hsc_oracle as orc
    from hsc_amd.modeling import LoCOMP
    x, D, kw = fz._draw(1353)
    kw = dict(kw, nbNonzeroCoefs=40)
    assert x.shape == (22,) and D.shape == (1, 14) and kw['nbBlocks'] == 8
    co, ro, _ = orc.cmp_encode(x, D, **kw)
    cd, rd = LoCOMP().computeCoefficients(x, D, **kw)
    ch, rh = LoCOMP(refit='host').computeCoefficients(x, D, **kw)
    assert float(abs(cd - co).max()) <= 1e-12 and float(abs(ch - co).max()) <= 1e-12
    assert float(np.max(np.abs(rd - ro))) <= 1e-12 and float(np.max(np.abs(rh - ro))) <= 1e-12
