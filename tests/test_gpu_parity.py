"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI (libhscmp.so via
hsc_amd), against (i) the CPU oracle on the same inputs -- bit-exact, indices AND floats, since
both pin the same fma chain and summation trees -- and (ii) the golden vectors of the real
reference: indices exact, floats within 1e-5 relative (float32) / 1e-10 (float64)."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-10}


def _oracle():
    from oracle import hsc_oracle as orc
    return orc


def _cmp():
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    return ConvolutionalMatchingPursuit()


def _assert_same_as_oracle(x, D, kw, got_coef, got_res, got_events, stats_row):
    from hsc_amd import _native
    orc = _oracle()
    coef, res, info = orc.cmp_encode(x, D, **kw)
    t, k, c = got_events
    assert np.array_equal(t, info['t']), 'positions differ from the oracle'
    assert np.array_equal(k, info['k']), 'atom indices differ from the oracle'
    assert np.array_equal(c, info['c']), 'coefficients are not bit-identical to the oracle'
    assert np.array_equal(got_res, res), 'residual is not bit-identical to the oracle'
    a, b = gu.csc_triplets(got_coef), gu.csc_triplets(coef)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    assert _native.STOP_NAMES[int(stats_row[_native.STAT_STOP])] == info['stop']
    assert int(stats_row[_native.STAT_NNZ]) == info['nnz']
    assert int(stats_row[_native.STAT_ROUNDS]) == info['rounds']


@pytest.mark.parametrize('name', gu.small_case_names())
def test_small_cases_vs_oracle_and_reference(name):
    x, D, kw, exp = gu.small_case(name)
    cmp = _cmp()
    coefficients, residual = cmp.computeCoefficients(x, D, **kw)
    res = cmp.lastResult
    _assert_same_as_oracle(x, D, kw, coefficients, residual, res.events[0], res.stats[0])
    # against the real reference (golden)
    from test_oracle_golden import NOISE_DRIVEN_STOP
    if name in NOISE_DRIVEN_STOP:
        return
    tol = TOL[np.result_type(x.dtype, D.dtype)]
    t, k, c = res.events[0]
    assert np.array_equal(t, exp['t']) and np.array_equal(k, exp['k'])
    assert gu.rel_err(c, exp['c']) <= tol
    row, col, data = gu.csc_triplets(coefficients)
    assert np.array_equal(row, exp['row']) and np.array_equal(col, exp['col'])
    assert gu.rel_err(data, exp['data']) <= tol
    assert residual.shape == exp['residual'].shape and residual.dtype == exp['residual'].dtype


@pytest.mark.parametrize('name', [str(n) for n in gu.load('functions.npz')['names']])
def test_convolve1d_vs_oracle_and_reference(name):
    from hsc_amd.modeling import convolve1d
    orc = _oracle()
    z = gu.load('functions.npz')
    x, D = z[name + '__x'], z[name + '__D']
    for padding in ('same', 'valid'):
        got = convolve1d(x, D, padding=padding)
        assert np.array_equal(got, orc.convolve1d(x, D, padding=padding))
        exp = z[name + '__' + padding]
        assert got.shape == exp.shape and got.dtype == exp.dtype
        assert float(np.max(np.abs(got.astype(np.float64) - exp))) <= 20 * TOL[x.dtype]


def test_batch_matches_single():
    """Signals of a batch are independent: batch results == one-by-one oracle results."""
    import hsc_amd.synth as synth
    D = synth.make_dictionary(32, 32, seed=1)
    xs = synth.make_batch(D, 4096, 0, 6, kind='planted', nb_atoms=64, seed=1)
    cmp = _cmp()
    res = cmp.computeCoefficientsBatch(xs, D, nbNonzeroCoefs=64)
    orc = _oracle()
    for b in range(xs.shape[0]):
        coef, r, info = orc.cmp_encode(xs[b], D, nbNonzeroCoefs=64)
        t, k, c = res.events[b]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c'])
        assert np.array_equal(res.residuals[b], r)


@pytest.mark.parametrize('kind', ['planted', 'noise'])
def test_config1_vs_reference_golden(kind):
    import hsc_amd.synth as synth
    z = gu.load('cmp_config.npz')
    D = synth.make_dictionary(32, 32, seed=1)
    x = synth.make_signal(D, 4096, 0, kind=kind, nb_atoms=64, seed=1)
    name = 'config1_%s' % kind
    assert synth.digest(x) == str(z[name + '__x_digest'])
    cmp = _cmp()
    coefficients, residual = cmp.computeCoefficients(x, D, nbNonzeroCoefs=64)
    t, k, c = cmp.lastResult.events[0]
    assert np.array_equal(t, z[name + '__t']) and np.array_equal(k, z[name + '__k'])
    assert gu.rel_err(c, z[name + '__c']) <= 1e-5
    e = float(np.sum(np.square(residual.astype(np.float64))))
    assert abs(e - float(z[name + '__residual_energy'])) <= 1e-5 * float(z[name + '__residual_energy'])


@pytest.mark.parametrize('kind,idx', [('planted', 0), ('planted', 1), ('planted', 2), ('planted', 3),
                                      ('noise', 0), ('noise', 1), ('noise', 2), ('noise', 3)])
def test_config2_full_size_vs_reference_golden(kind, idx):
    """BASELINE config 2 shape (T=65536, K=256, W=64, L0=256): 8 signals encoded by the real
    reference in the build container; indices exact, coefficients 1e-5, residual energy 1e-5."""
    import hsc_amd.synth as synth
    z = gu.load('cmp_config.npz')
    D = synth.make_dictionary(256, 64, seed=2)
    x = synth.make_signal(D, 65536, idx, kind=kind, nb_atoms=256, seed=2)
    name = 'config2_%s_%d' % (kind, idx)
    assert synth.digest(x) == str(z[name + '__x_digest'])
    cmp = _cmp()
    coefficients, residual = cmp.computeCoefficients(x, D, nbNonzeroCoefs=256)
    t, k, c = cmp.lastResult.events[0]
    assert np.array_equal(t, z[name + '__t']) and np.array_equal(k, z[name + '__k'])
    assert gu.rel_err(c, z[name + '__c']) <= 1e-5
    row, col, data = gu.csc_triplets(coefficients)
    assert np.array_equal(row, z[name + '__csc_row']) and np.array_equal(col, z[name + '__csc_col'])
    assert gu.rel_err(data, z[name + '__csc_data']) <= 1e-5
    e = float(np.sum(np.square(residual.astype(np.float64))))
    assert abs(e - float(z[name + '__residual_energy'])) <= 1e-5 * float(z[name + '__residual_energy'])


@pytest.mark.parametrize('name', [str(n) for n in gu.load('functions.npz')['names']])
def test_select_best_atoms_entry_point_vs_reference(name):
    """Row a2 on its own: _selectBestAtoms (modeling.py:899-982) through hscmp_select_best_atoms on the
    reference's table, 9 modes per case (single / blocked / 'auto', offsets, weights): exact."""
    z = gu.load('functions.npz')
    ip = z[name + '__same']
    W = z[name + '__D'].shape[1]
    cmp = _cmp()
    for i in range(int(z[name + '__nsel'])):
        nb = int(z['%s__sel%d_nb' % (name, i)])
        off = bool(int(z['%s__sel%d_offset' % (name, i)]))
        wkey = '%s__sel%d_weights' % (name, i)
        w = z[wkey] if wkey in z else None
        atoms = cmp._selectBestAtoms(ip, W, nbBlocks='auto' if nb == -1 else nb, offset=off, nullCoeffThres=1e-16, weights=w)
        assert [a.position for a in atoms] == z['%s__sel%d_t' % (name, i)].tolist()
        assert [a.index for a in atoms] == z['%s__sel%d_k' % (name, i)].tolist()
        assert np.array_equal(np.array([a.coefficient for a in atoms], dtype=ip.dtype), z['%s__sel%d_c' % (name, i)])
        assert all(a.length == W for a in atoms)


def test_select_best_atoms_reference_kat():
    """tests/hsc/test_modeling.py:272-325 through the GPU entry point."""
    cmp = _cmp()
    ip = np.arange(256).reshape((64, 4)).astype(np.float64)
    ip[-1] = ip[-1][::-1]
    pos = lambda atoms: ([a.position for a in atoms], [a.index for a in atoms])
    assert pos(cmp._selectBestAtoms(ip, 5, 4, offset=False)) == ([63, 47, 31, 15], [0, 3, 3, 3])
    assert pos(cmp._selectBestAtoms(ip, 5, 4, offset=True)) == ([63, 55, 39, 23, 7], [0, 3, 3, 3, 3])
    assert pos(cmp._selectBestAtoms(ip, 3, 'auto', offset=False)) == ([63, 59, 47, 35, 23, 11], [0, 3, 3, 3, 3, 3])
    assert pos(cmp._selectBestAtoms(ip, 5, 5, offset=False)) == ([59, 47, 35, 23, 11], [3, 3, 3, 3, 3])


@pytest.mark.parametrize('name', [str(n) for n in gu.load('functions.npz')['names']])
def test_update_inner_products_entry_point_vs_reference(name):
    """Row a5 on its own: _updateInnerProducts (modeling.py:1018-1051) incl. the reflect-padding quirk at
    both edges, through hscmp_update_inner_products: bit-exact vs the oracle, 1e-5 vs the reference."""
    from hsc_amd.modeling import Atom
    orc = _oracle()
    z = gu.load('functions.npz')
    D = z[name + '__D']
    ip = z[name + '__same'].copy()
    ipo = ip.copy()
    cmp = _cmp()
    for j in range(int(z[name + '__nupd'])):
        p = int(z['%s__upd%d_p' % (name, j)])
        r = z['%s__upd%d_r' % (name, j)]
        cmp._updateInnerProducts(ip, r, [Atom(p, 0, 1.0, D.shape[1])], D)
        orc.update_inner_products(ipo, r, D, p)
        assert np.array_equal(ip, ipo)
        assert float(np.max(np.abs(ip.astype(np.float64) - z['%s__upd%d_ip' % (name, j)]))) <= 20 * TOL[ip.dtype]
