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
    """Mixin replacing the GPU-backed hooks by the CPU oracle."""

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


@pytest.mark.gpu
@pytest.mark.parametrize('name', _names())
def test_locomp_gpu_vs_reference_golden(name):
    from hsc_amd.modeling import LoCOMP
    _check(LoCOMP(), name)


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
