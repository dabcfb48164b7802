"""The reference's DEFAULT hierarchical method (hsc/modeling.py:1429 method='locomp') and LoCOMP's re-fit on rank-deficient groups,
against golden vectors of the real reference (tests/golden/locomp_hier.npz, tools/make_golden.py locomphier).

Every fixture carries what the reference's np.linalg.pinv (:1326) saw while it produced it: the number of groups, the largest one, how
many groups had a singular value cut (rcond 1e-15 in the dictionary's dtype) and the smallest KEPT singular value relative to the
largest, i.e. 1 / condition of the worst group.  The tolerances below are stated from that: the reference's own SVD runs in the
dictionary's dtype, so its coefficients carry eps(dtype) x condition of round-off; the engine re-fits in float64.

CPU tests run the host loop with the CPU oracle behind the three GPU hooks (they pin the fixtures and the host logic); the -m gpu tests
run the BATCH DEVICE path -- LoCOMP().computeCoefficientsBatch / HierarchicalConvolutionalMatchingPursuit('locomp').
computeCoefficientsBatch, the loop of csrc/hscmp_locomp.h -- which is what bench.py --method locomp times."""
import hashlib
import os

import numpy as np
import pytest

import golden_util as gu


def _z():
    return gu.load('locomp_hier.npz')


def _single_names():
    return [str(n) for n in _z()['names'] if str(n).startswith(('rd_', 'soak_'))]


def _single_case(name):
    z = _z()
    kw = {}
    for key in ('nbNonzeroCoefs', 'toleranceSnr', 'minCoefficients', 'toleranceResidualScale'):
        full = '%s__%s' % (name, key)
        if full in z:
            kw[key] = int(z[full]) if key == 'nbNonzeroCoefs' else float(z[full])
        if full + '_is_none' in z:
            kw[key] = None
    if name + '__nbBlocks' in z:
        nb = int(z[name + '__nbBlocks'])
        kw['nbBlocks'] = 'auto' if nb == -1 else nb
    if name + '__weights' in z:
        kw['weights'] = z[name + '__weights']
    exp = dict(residual=z[name + '__residual'], row=z[name + '__csc_row'], col=z[name + '__csc_col'], data=z[name + '__csc_data'],
               cond=1.0 / float(z[name + '__min_rel_sigma_kept']), cut=int(z[name + '__groups_cut']))
    return z[name + '__x'], z[name + '__D'], kw, exp


def _tolerance(dtype, cond):
    """Coefficients of the reference carry eps(dictionary dtype) x condition of the worst group (its SVD runs in that dtype); 50 x
    that, and never below the 1e-5 (float32) / 1e-9 (float64) of the greedy coder's comparisons."""
    if np.dtype(dtype) == np.float32:
        return max(1e-5, 50 * 1.2e-7 * cond)
    return max(1e-9, 50 * 2.3e-16 * cond)


def _without_roundoff_atoms(row, col, data):
    """Entries below 1e-12 of the largest coefficient: atoms fitted to a residual that is already round-off (soak_117: the group fills
    the whole 15-sample signal, the re-fit reproduces it exactly, and whether the loop then runs once more -- energy 1e-60 against eps --
    is decided by the last bit; the reference's extra entry is 2e-16)."""
    keep = np.abs(data) > 1e-12 * max(1e-300, float(np.max(np.abs(data))) if len(data) else 0.0)
    return row[keep], col[keep], data[keep]


def _check_single(coefficients, residual, x, D, exp):
    row, col, data = _without_roundoff_atoms(*gu.csc_triplets(coefficients))
    exp = dict(exp)
    exp['row'], exp['col'], exp['data'] = _without_roundoff_atoms(exp['row'], exp['col'], exp['data'])
    assert np.array_equal(row, exp['row']) and np.array_equal(col, exp['col']), 'support differs from the reference'
    tol = _tolerance(np.result_type(x.dtype, D.dtype), exp['cond'])
    scale = max(1.0, float(np.max(np.abs(exp['data']))))
    assert float(np.max(np.abs(data - exp['data']))) <= tol * scale, (float(np.max(np.abs(data - exp['data']))), tol)
    assert residual.shape == exp['residual'].shape and residual.dtype == exp['residual'].dtype
    assert float(np.max(np.abs(residual.astype(np.float64) - exp['residual']))) <= 10 * tol * scale


def test_fixture_inventory():
    """>= 8 single-level cases with a cut singular value (rank-deficient groups), the three tiny-signal soak draws among them, and the
    four hierarchical cases; the soak draws are the inputs tests/test_gpu_fuzz.py::_draw still produces."""
    import test_gpu_fuzz as fz
    z = _z()
    names = _single_names()
    assert len([n for n in names if int(z[n + '__groups_cut']) > 0]) >= 8
    for i in (29, 53, 117):
        x, D, _ = fz._draw(i)
        assert np.array_equal(x, z['soak_%d__x' % i]) and np.array_equal(D, z['soak_%d__D' % i])
    for n in ('c4w16', 'c4w17', 'gen3', 'gen3_f64'):
        assert int(z[n + '__groups']) > 0 and float(z[n + '__min_rel_sigma_kept']) > 0.1     # (well-conditioned groups throughout)


@pytest.mark.parametrize('name', _single_names())
def test_rank_deficient_groups_host_loop_with_oracle_hooks(name):
    """np.linalg.pinv on the host, the oracle behind the hooks: pins the fixtures (and the host loop that takes over from the kernel)."""
    import hsc_amd.locomp as locomp
    from test_locomp import _OracleHooks

    class OracleLoCOMP(_OracleHooks, locomp.LoCOMP):
        pass

    x, D, kw, exp = _single_case(name)
    coefficients, residual = OracleLoCOMP().computeCoefficients(x, D, **kw)
    _check_single(coefficients, residual, x, D, exp)


@pytest.mark.gpu
@pytest.mark.parametrize('name', _single_names())
def test_rank_deficient_groups_batch_device_path(name):
    """The kernel's re-fit (float64 normal equations; minimum-norm completion when a pivot vanishes) against the reference's
    pseudo-inverse on groups where it cut a singular value: support exact, coefficients within the stated tolerance, no signal
    handed to the host loop."""
    from hsc_amd.modeling import LoCOMP
    x, D, kw, exp = _single_case(name)
    assert exp['cut'] > 0
    coder = LoCOMP()
    res = coder.computeCoefficientsBatch(np.stack([x, x]), D, **kw)
    assert 'locomp' in res.variant and 'group' not in res.stop_reasons()
    for b in range(2):
        _check_single(res.coefficients[b], res.residuals[b], x, D, exp)


def _hier_inputs(name):
    """(multilevel dictionary with singleton bases, signal, encode kwargs) of a hierarchical fixture, regenerated from its seeds."""
    z = _z()
    if name.startswith('c4w'):
        import hsc_amd.synth as synth
        W1 = int(name[3:])
        own = synth.make_hierarchy(W1=W1, seed=4)
        x = synth.make_hierarchy_signal(own, 8192, 0, seed=4)
        assert synth.digest(x) == str(z[name + '__x_digest'])
        return own.withSingletonBases(), x, dict(toleranceSnr=[30.0, 40.0], nbBlocks=10, singletonWeight=0.95)
    import copy
    from hsc_amd.dataset import MultilevelDictionary, MultilevelDictionaryGenerator, SignalGenerator
    np.random.seed(77)
    mld = MultilevelDictionaryGenerator().generate(scales=[16, 32, 64], counts=[16, 24, 32], decompositionSize=3,
                                                   multilevelDecomposition=False, maxNbPatternsConsecutiveRejected=50)
    np.random.seed(78)
    gen = SignalGenerator(mld, [0.004, 0.004, 0.004])
    events = gen.generateEvents(4096)
    x = gen.generateSignalFromEvents(events, nbSamples=4096)
    if name.endswith('_f64'):
        dec64 = [[[d[0], d[1], d[2], np.asarray(d[3], dtype=np.float64)] for d in lev] for lev in copy.deepcopy(mld.decompositions)]
        mld = MultilevelDictionary.fromDecompositions(mld.dictionaries[0].astype(np.float64), dec64, mld.scales)
        assert all(d.dtype == np.float64 for d in mld.dictionaries)
        x = x.astype(np.float64)
    assert len(events) == int(z[name + '__nevents'])
    assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == str(z[name + '__x_sha256'])
    return mld.withSingletonBases(), x, dict(toleranceSnr=[20.0, 30.0, 30.0], nbBlocks=10, singletonWeight=0.9)


def _check_hier(name, coefficients, energy):
    z = _z()
    nlev = int(z[name + '__nlevels'])
    assert len(coefficients) == nlev
    # the level dictionaries are float32 in c4w* / gen3 (the reference's pinv then runs in float32 at EVERY level), float64 in gen3_f64
    tol = _tolerance(np.float64 if name.endswith('_f64') else np.float32, 1.0 / float(z[name + '__min_rel_sigma_kept']))
    for l, c in enumerate(coefficients):
        row, col, data = gu.csc_triplets(c)
        assert np.array_equal(row, z['%s__level%d_row' % (name, l)]) and np.array_equal(col, z['%s__level%d_col' % (name, l)]), (name, l)
        exp = z['%s__level%d_data' % (name, l)]
        scale = max(1.0, float(np.max(np.abs(exp))) if len(exp) else 1.0)
        assert (float(np.max(np.abs(data - exp))) if len(exp) else 0.0) <= tol * scale, (name, l, float(np.max(np.abs(data - exp))), tol)
    e = float(z[name + '__residual_energy'])
    assert abs(energy - e) <= 1e-4 * e, (energy, e)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['c4w16', 'c4w17', 'gen3', 'gen3_f64'])
def test_hierarchical_locomp_batch_device_path_vs_reference(name):
    """HierarchicalConvolutionalMatchingPursuit(method='locomp').computeCoefficientsBatch -- every level on the LoCOMP loop of
    csrc/hscmp_locomp.h, device-chained levels, device epilogue -- against the reference's encode with its default method: per level
    the support exact, the coefficients within the stated tolerance; residual energy 1e-4.  The batch holds the signal three times:
    all three results are identical."""
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    mlds, x, kw = _hier_inputs(name)
    h = HierarchicalConvolutionalMatchingPursuit(method='locomp')
    try:
        coefs, energies, timings = h.computeCoefficientsBatch(np.stack([x, x, x]), mlds, residuals='energy', **kw)
    finally:
        h.close()
    assert all('locomp' in tm['variant'] for tm in timings)
    for b in range(3):
        _check_hier(name, coefs[b], float(energies[b]))
        for l in range(len(coefs[0])):
            assert (coefs[b][l] != coefs[0][l]).nnz == 0


@pytest.mark.gpu
def test_group_capacity_override_forces_the_host_loop_even_with_a_placeholder_host_array(monkeypatch):
    """HSCMP_LOCOMP_GROUP_CAP=3: the kernel gives up on every signal whose groups exceed two neighbours (stop reason 'group'), the
    batch entry repeats those signals through the per-signal entry -- and with deviceInput set it must read them back from the device:
    the host array passed alongside is zeros (shape and dtype only)."""
    import torch
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    mlds, x, kw = _hier_inputs('c4w17')
    xs = np.stack([x, 0.5 * x]).astype(np.float32)
    h = HierarchicalConvolutionalMatchingPursuit(method='locomp')
    try:
        want, want_e, _ = h.computeCoefficientsBatch(xs, mlds, residuals='energy', **kw)
        monkeypatch.setenv('HSCMP_LOCOMP_GROUP_CAP', '3')
        xd = torch.from_numpy(xs).cuda()
        got, got_e, _ = h.computeCoefficientsBatch(np.zeros_like(xs), mlds, residuals='energy', deviceInput=xd.data_ptr(), **kw)
    finally:
        h.close()
    for b in range(2):
        for l in range(2):
            a, w = got[b][l].tocsc(), want[b][l].tocsc()
            assert a.nnz == w.nnz and np.array_equal(a.indices, w.indices) and np.array_equal(a.indptr, w.indptr), (b, l)
            assert float(np.max(np.abs(a.data - w.data))) <= 1e-4
        assert abs(got_e[b] - want_e[b]) <= 1e-3 * want_e[b]
