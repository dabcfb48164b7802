"""Hierarchical encoder (hsc/modeling.py:1427-1705) and the MultilevelDictionary container
(hsc/dataset.py:110-410) against golden vectors of the real reference (tests/golden/hsc_small.npz).

CPU tests run the per-level loop with the level coder replaced by the CPU oracle (test
infrastructure) -- they check the host logic: weights, per-level SNR, densification, distributed
post-processing, residual.  The -m gpu tests run the product path (GPU engine)."""
import os

import numpy as np
import pytest
import scipy.sparse

import golden_util as gu


def _golden():
    return gu.load('hsc_small.npz')


def _mld():
    from hsc_amd.dataset import MultilevelDictionary
    z = _golden()
    dicts = [z['raw0'], z['raw1'], z['raw2']]
    return MultilevelDictionary.fromRawDictionaries(dicts, [int(s) for s in z['scales']])


def _mld_first_level(mld):
    from hsc_amd.dataset import MultilevelDictionary
    return MultilevelDictionary.fromRawDictionaries([mld.getRawDictionary(0)], [int(_golden()['scales'][0])])


def test_multilevel_dictionary_matches_reference():
    z = _golden()
    mld = _mld()
    for l in range(3):
        assert np.allclose(mld.getMultiscaleDictionaries()[l], z['rep%d' % l], atol=1e-7)
    mlds = mld.withSingletonBases()
    assert mlds.hasSingletonBases
    assert np.array_equal(mlds.counts, z['counts'])
    assert np.array_equal(mlds.countsNoSingletons, z['countsNoSingletons'])
    for l in range(3):
        assert np.array_equal(mlds.getRawDictionary(l), z['single_raw%d' % l])
        assert np.allclose(mlds.getMultiscaleDictionaries()[l], z['single_rep%d' % l], atol=1e-7)
    assert mlds.withSingletonBases() is mlds


def _case_kwargs(z, name):
    snr = z['case_%s__toleranceSnr' % name]
    kw = dict(toleranceSnr=[float(v) for v in snr] if int(z['case_%s__snr_is_list' % name]) else float(snr[0]),
              singletonWeight=float(z['case_%s__singletonWeight' % name]),
              returnDistributed=bool(int(z['case_%s__returnDistributed' % name])))
    nb = int(z['case_%s__nbBlocks' % name])
    kw['nbBlocks'] = 'auto' if nb == -1 else nb
    return kw


def _check_case(hcsc, z, name):
    x = z['x']
    kw = _case_kwargs(z, name)
    coefficients, residual = hcsc.encode(x, **kw)
    assert len(coefficients) == 3
    for l, c in enumerate(coefficients):
        row, col, data = gu.csc_triplets(c)
        assert np.array_equal(row, z['case_%s__level%d_row' % (name, l)]), 'level %d structure' % l
        assert np.array_equal(col, z['case_%s__level%d_col' % (name, l)])
        assert gu.rel_err(data, z['case_%s__level%d_data' % (name, l)]) <= 1e-5
    exp_res = z['case_%s__residual' % name]
    assert residual.shape == exp_res.shape and residual.dtype == exp_res.dtype
    assert float(np.max(np.abs(residual - exp_res))) <= 1e-5
    if kw['returnDistributed']:
        assert float(np.max(np.abs(residual))) < float(np.max(np.abs(x)))      # tests/hsc/test_modeling.py:823-869
    assert float(np.max(np.abs(hcsc.reconstruct(coefficients) - z['case_%s__recon' % name]))) <= 1e-5
    # encodeFromLevel: resume above level 0 (modeling.py:1645-1654)
    raw = hcsc.approximator._forwardPhase(x, hcsc.multilevelDict, kw['toleranceSnr'], kw['nbBlocks'], kw['singletonWeight'])
    cont = hcsc.encodeFromLevel(x, raw[:1], toleranceSnr=kw['toleranceSnr'], nbBlocks=kw['nbBlocks'],
                                singletonWeight=kw['singletonWeight'], returnDistributed=kw['returnDistributed'])
    for l, c in enumerate(cont):
        row, col, data = gu.csc_triplets(c)
        assert np.array_equal(row, z['case_%s__fromlevel%d_row' % (name, l)])
        assert np.array_equal(col, z['case_%s__fromlevel%d_col' % (name, l)])
        assert gu.rel_err(data, z['case_%s__fromlevel%d_data' % (name, l)]) <= 1e-5


class _OracleLevelCoder(object):
    """Stand-in level coder for the CPU tests: same encode() contract, computed by the CPU oracle."""

    def __init__(self, D):
        self.D = D

    def encode(self, X, **kw):
        from oracle import hsc_oracle as orc
        coefficients, residual, _ = orc.cmp_encode(np.asarray(X), self.D, **kw)
        return coefficients, residual


@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_hierarchical_host_logic_with_oracle_level_coder(name, monkeypatch):
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit, HierarchicalConvolutionalSparseCoder
    hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    monkeypatch.setattr(hcmp, '_level_coder', lambda D: _OracleLevelCoder(D))
    hcsc = HierarchicalConvolutionalSparseCoder(_mld(), hcmp)
    _check_case(hcsc, _golden(), name)


def test_unsupported_methods_raise():
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    x = _golden()['x']
    mld = _mld().withSingletonBases()
    with pytest.raises(NotImplementedError):
        HierarchicalConvolutionalMatchingPursuit(method='mptk-cmp').computeCoefficients(x, mld, toleranceSnr=5.0)
    with pytest.raises(Exception):
        HierarchicalConvolutionalMatchingPursuit(method='nope').computeCoefficients(x, mld, toleranceSnr=5.0)


def test_distributed_conversion_preserves_nnz():
    """modeling.py:1556-1594 on a hand-made example."""
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    c0 = scipy.sparse.csc_matrix((6, 2))
    c1 = scipy.sparse.csc_matrix(np.array([[1., 0, 0, 2.], [0, 0, 0, 0], [0, 3., 4., 0], [0, 0, 0, 0], [0, 0, 5., 0], [0, 0, 0, 0]]))
    out = HierarchicalConvolutionalMatchingPursuit('cmp').convertToDistributedCoefficients([c0, c1])
    assert out[0].shape == (6, 2) and out[0].nnz == 2          # the two singleton columns went down a level
    assert out[1].shape == (6, 4) and out[1].nnz == 3
    assert out[1][:, :2].nnz == 0


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_hierarchical_gpu_vs_reference_golden(name):
    from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit, HierarchicalConvolutionalSparseCoder
    hcsc = HierarchicalConvolutionalSparseCoder(_mld(), HierarchicalConvolutionalMatchingPursuit(method='cmp'))
    _check_case(hcsc, _golden(), name)


@pytest.mark.gpu
def test_hierarchical_batch_equals_per_signal():
    from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit
    z = _golden()
    mld = _mld().withSingletonBases()
    rs = np.random.RandomState(9)
    xs = np.stack([z['x'], (z['x'][::-1]).copy(), (z['x'] * 0.5 + 0.02 * rs.standard_normal(z['x'].shape)).astype(np.float32)])
    hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    kw = dict(toleranceSnr=[15.0, 20.0, 20.0], nbBlocks=4, singletonWeight=0.9)
    for chained, budget in ((True, 64e9), (True, 2.5e6), (False, 64e9)):     # 2.5 MB: forces 1-2 signals per chunk
        coefs, residuals, timings = hcmp.computeCoefficientsBatch(xs, mld, chained=chained, memoryBudget=budget, **kw)
        assert len(timings) == 3 and timings[1]['variant'].startswith('dictlist_init+dictlist_loop')
        for b in range(xs.shape[0]):
            c1, r1 = hcmp.computeCoefficients(xs[b], mld, **kw)
            for l in range(3):
                assert (coefs[b][l] != c1[l]).nnz == 0
            assert np.array_equal(residuals[b], r1)


def test_events_wire_format_matches_reference():
    """hsc/dataset.py:798-824: coefficient matrices <-> (time, level, index, coefficient) records."""
    from hsc_amd.dataset import convertSparseMatricesToEvents, convertEventsToSparseMatrices
    z = _golden()
    counts = [z['single_raw%d' % l].shape[0] for l in range(3)]
    coefs = [scipy.sparse.csc_matrix((z['case_a__level%d_data' % l], (z['case_a__level%d_row' % l], z['case_a__level%d_col' % l])),
                                     shape=(512, counts[l])) for l in range(3)]
    ev = convertSparseMatricesToEvents(coefs)
    assert ev.dtype == np.dtype('int32,int32,int32,float32')
    assert np.array_equal(ev['f0'], z['case_a__events_t']) and np.array_equal(ev['f1'], z['case_a__events_l'])
    assert np.array_equal(ev['f2'], z['case_a__events_i']) and np.array_equal(ev['f3'], z['case_a__events_c'])
    back = convertEventsToSparseMatrices(ev, counts, 512)
    for l, b in enumerate(back):
        row, col, data = gu.csc_triplets(b)
        assert np.array_equal(row, z['case_a__back%d_row' % l]) and np.array_equal(col, z['case_a__back%d_col' % l])
        assert np.array_equal(data.astype(np.float32), z['case_a__back%d_data' % l].astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['a', 'b'])
def test_hierarchical_medium_generated_data_vs_reference(name):
    """2-level encode of generated data (Perlin dictionary, Poisson events, 8192 samples; level-1 dictionary
    (24+20) x 33 x 24, sparse x sparse kernels) against the coefficients of the REAL reference
    (tests/golden/hsc_medium.npz, tools/make_golden.py hscmed): positions and atoms exact, values 1e-5."""
    import hashlib
    from hsc_amd.dataset import MultilevelDictionaryGenerator, SignalGenerator
    from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit, HierarchicalConvolutionalSparseCoder
    z = gu.load('hsc_medium.npz')
    np.random.seed(77)
    mld = MultilevelDictionaryGenerator().generate(scales=[32, 64], counts=[24, 20], decompositionSize=3,
                                                   multilevelDecomposition=False, maxNbPatternsConsecutiveRejected=50)
    np.random.seed(78)
    gen = SignalGenerator(mld, [0.004, 0.004])
    events = gen.generateEvents(8192)
    x = gen.generateSignalFromEvents(events, nbSamples=8192)
    assert len(events) == int(z['nevents'])
    assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == str(z['x_sha256'])
    kw = dict(a=dict(toleranceSnr=[20.0, 25.0], nbBlocks=8, singletonWeight=0.9),
              b=dict(toleranceSnr=[25.0, 30.0], nbBlocks='auto', singletonWeight=0.95, returnDistributed=False))[name]
    hcsc = HierarchicalConvolutionalSparseCoder(mld, HierarchicalConvolutionalMatchingPursuit(method='cmp'))
    coefficients, residual = hcsc.encode(x, **kw)
    for l, c in enumerate(coefficients):
        row, col, data = gu.csc_triplets(scipy.sparse.csc_matrix(c))
        assert np.array_equal(row, z['case_%s__level%d_row' % (name, l)]), l
        assert np.array_equal(col, z['case_%s__level%d_col' % (name, l)]), l
        assert gu.rel_err(data, z['case_%s__level%d_data' % (name, l)]) <= 1e-5
    e = float(np.sum(np.square(np.asarray(residual, dtype=np.float64))))
    assert abs(e - float(z['case_%s__residual_energy' % name])) <= 1e-5 * float(z['case_%s__residual_energy' % name])


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(int(os.environ.get('HSCMP_FUZZ_HSC', '24'))))
def test_hierarchical_random_generated_cases_gpu_equals_oracle_level_coder(seed, monkeypatch):
    """Random generated dictionaries / signals / parameters: the GPU hierarchical encoder (per-signal path and the
    device-chained batch path) against the same host logic driven by the CPU oracle as its level coder -- bit for
    bit, all levels."""
    from hsc_amd.dataset import MultilevelDictionaryGenerator, SignalGenerator
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    rs = np.random.RandomState(500 + seed)
    nlev = 2 + (seed % 3)
    scales = [int(rs.choice([8, 12, 16]))]
    for _ in range(nlev - 1):
        scales.append(scales[-1] * 2 + int(rs.randint(0, 5)))
    counts = [int(rs.randint(4, 9)) for _ in range(nlev)]
    mld = MultilevelDictionaryGenerator(rs).generate(scales=scales, counts=counts, decompositionSize=int(rs.randint(2, 4)),
                                                     multilevelDecomposition=False, maxNbPatternsConsecutiveRejected=30)
    T = int(rs.randint(20, 60)) * scales[-1]
    gen = SignalGenerator(mld, [0.02 / (l + 1) for l in range(nlev)], rng=rs)
    xs = np.stack([gen.generateSignalFromEvents(gen.generateEvents(T), nbSamples=T) for _ in range(3)])
    xs = (xs + 0.01 * rs.standard_normal(xs.shape)).astype(np.float32)
    mlds = mld.withSingletonBases()
    kw = dict(toleranceSnr=[float(rs.uniform(8, 16)) for _ in range(nlev)], nbBlocks=[1, 3, 'auto'][seed % 3],
              singletonWeight=float(rs.uniform(0.8, 0.95)))
    gpu = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    ref = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    monkeypatch.setattr(ref, '_level_coder', lambda D: _OracleLevelCoder(D))
    from hsc_amd._native import HscmpError
    try:
        coefs_b, residuals_b, timings = gpu.computeCoefficientsBatch(xs, mlds, **kw)
    except HscmpError as ex:
        # a level whose pursuit does not terminate under the drawn parameters (the reference would loop forever):
        # the oracle-driven run must hit its own event capacity on that input as well
        assert 'does not converge' in str(ex)
        pytest.skip('non-terminating random case')
    for b in range(xs.shape[0]):
        exp_c, exp_r = ref.computeCoefficients(xs[b], mlds, **kw)
        got_c, got_r = gpu.computeCoefficients(xs[b], mlds, **kw)
        for l in range(nlev):
            assert (scipy.sparse.csc_matrix(got_c[l]) != scipy.sparse.csc_matrix(exp_c[l])).nnz == 0, (seed, b, l, 'per-signal')
            assert (scipy.sparse.csc_matrix(coefs_b[b][l]) != scipy.sparse.csc_matrix(exp_c[l])).nnz == 0, (seed, b, l, 'batch')
        assert np.array_equal(got_r, exp_r) and np.array_equal(residuals_b[b], exp_r)
    # the signal energy of a chained level through the counting sort of its input slots (csrc: prepare_from_slots_sorted_kernel,
    # dispatched for long slot lists only) and without it: the same stop decisions, the same everything
    for key, val in (('HSCMP_SORTED_PREPARE_MIN', '0'), ('HSCMP_NO_SORTED_PREPARE', '1')):
        monkeypatch.setenv(key, val)
        coefs_s, residuals_s, _ = gpu.computeCoefficientsBatch(xs, mlds, **kw)
        monkeypatch.delenv(key)
        assert np.array_equal(residuals_s, residuals_b), (seed, key)
        for b in range(xs.shape[0]):
            for l in range(nlev):
                assert (scipy.sparse.csc_matrix(coefs_s[b][l]) != scipy.sparse.csc_matrix(coefs_b[b][l])).nnz == 0, (seed, b, l, key)
    gpu.close()


@pytest.mark.gpu
def test_hierarchical_batch_with_locomp_runs_the_device_loop_on_every_level():
    """The batch entry point with the reference's default method ('locomp'): the chained pipeline of method='cmp' with every
    level's engine on the LoCOMP loop (csrc/hscmp_locomp.h).  Deterministic, the same from the device and the host epilogue,
    the residual is the signal minus the reconstruction of the returned code, and level 0 equals the stand-alone batch LoCOMP.
    (Against the per-signal entry -- the reference's own pseudo-inverse on the host -- see the comment at the end.)"""
    from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit, LoCOMP
    z = _golden()
    mld = _mld().withSingletonBases()
    rs = np.random.RandomState(2)
    xs = np.stack([z['x'], (z['x'][::-1]).copy()] + [np.roll(z['x'], int(s)) * np.float32(a) for s, a in zip(rs.randint(1, 200, 10), rs.uniform(0.5, 2.0, 10))])
    hcmp = HierarchicalConvolutionalMatchingPursuit(method='locomp')
    kw = dict(toleranceSnr=[10.0, 20.0, 20.0], nbBlocks=4, singletonWeight=0.5)
    coefs, residuals, timings = hcmp.computeCoefficientsBatch(xs, mld, **kw)
    assert all('locomp' in t['variant'] for t in timings)
    coefs2, residuals2, _ = hcmp.computeCoefficientsBatch(xs, mld, **kw)
    coefs3, residuals3, _ = hcmp.computeCoefficientsBatch(xs, mld, epilogue='host', **kw)
    assert np.array_equal(residuals, residuals2) and np.array_equal(residuals, residuals3)
    for b in range(xs.shape[0]):
        for l in range(3):
            assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(coefs2[b][l])).nnz == 0
            assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(coefs3[b][l])).nnz == 0
        r = hcmp._calculateResidual(xs[b], coefs[b], mld)
        assert np.allclose(residuals[b], r, rtol=0.0, atol=1e-9)
    # level 0 alone (not distributed: the last level's singleton columns are what returnDistributed hands back)
    first, _, _ = hcmp.computeCoefficientsBatch(xs, _mld_first_level(mld), toleranceSnr=[10.0], nbBlocks=4, singletonWeight=0.5, returnDistributed=False)
    D0 = mld.getRawDictionary(0)
    alone = LoCOMP().computeCoefficientsBatch(xs, D0, toleranceSnr=10.0, nbBlocks=4, weights=np.ones((D0.shape[0],), dtype=D0.dtype))
    for b in range(xs.shape[0]):
        assert (scipy.sparse.csc_matrix(first[b][0]) != alone.coefficients[b]).nnz == 0
    # Against the host loop (np.linalg.pinv in the dictionary's dtype, as the reference): level by level ON THE SAME INPUT -- the
    # host loop's own output of the level below -- so that no cascade is involved.  What remains between the two is (a) entries
    # that are exactly 0.0 in the float64 re-fit and ~1e-8 in the reference's float32 pseudo-inverse (they count in nnz, not in
    # value), (b) stops decided by round-off ('stalled': |dE| < eps of the float32 dictionary, modeling.py:1379), after which one of
    # the two runs a few atoms longer.  Values agree wherever both have an entry of size; the reference goldens of the batch path
    # (tests/test_locomp_hier.py) are the tight comparison.
    snr = kw['toleranceSnr']
    for b in range(0, xs.shape[0], 3):
        inp = xs[b]
        for level in range(3):
            D = mld.getRawDictionary(level)
            w = np.ones(D.shape[0], dtype=D.dtype); w[:D.shape[0] - mld.countsNoSingletons[level]] = 0.5
            res = LoCOMP().computeCoefficientsBatch(np.asarray(inp)[None], D, toleranceSnr=snr[level], nbBlocks=4, weights=w)
            ch, _ = LoCOMP(refit='host').computeCoefficients(np.asarray(inp), D, toleranceSnr=snr[level], nbBlocks=4, weights=w)
            a, h = res.coefficients[0].tocsc(), ch.tocsc()
            big_a, big_h = int((abs(a.data) > 1e-6).sum()), int((abs(h.data) > 1e-6).sum())
            assert abs(big_a - big_h) <= max(3, big_h // 8), (b, level, big_a, big_h)
            assert float(abs(a - h).max()) <= 0.05 * float(abs(h).max()), (b, level)
            inp = ch.toarray()
    hcmp.close()


@pytest.mark.gpu
@pytest.mark.parametrize('lds_keys', [None, '64'])
@pytest.mark.parametrize('distributed', [True, False])
def test_device_epilogue_equals_host_epilogue(distributed, lds_keys, monkeypatch):
    """hscmp_hierarchy_epilogue (redistribution, CSC, events, residual on the device) against the host epilogue
    (hscmp_host_slots_to_csc + scipy column slicing + hscmp_host_overlap_add): coefficient matrices, event records and
    the float64 residual bit for bit; 3 levels, several chunks.  HSCMP_EPI_LDS_KEYS=64: the slot lists do not fit the
    key buffer in LDS -- the sorts run chunk by chunk through global memory (as for more than 16384 slots per signal), and the
    residual tiles are small enough that some are denser than their entry list (the per-sample form)."""
    if lds_keys is not None:
        monkeypatch.setenv('HSCMP_EPI_LDS_KEYS', lds_keys)
    from hsc_amd.dataset import convertSparseMatricesToEvents
    from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit
    z = _golden()
    mld = _mld().withSingletonBases()
    rs = np.random.RandomState(19)
    base = z['x']
    xs = np.stack([base, base[::-1].copy(), (0.7 * base + 0.05 * rs.standard_normal(base.shape)).astype(np.float32),
                   (base * np.float32(1.3)), np.roll(base, 37)])
    hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    kw = dict(toleranceSnr=[15.0, 20.0, 20.0], nbBlocks=4, singletonWeight=0.9, returnDistributed=distributed)
    for budget in (64e9, 2.5e6):
        cd, rd, _, ed = hcmp.computeCoefficientsBatch(xs, mld, memoryBudget=budget, epilogue='device', returnEvents=True, **kw)
        ch, rh, _, eh = hcmp.computeCoefficientsBatch(xs, mld, memoryBudget=budget, epilogue='host', returnEvents=True, **kw)
        assert rd.shape == rh.shape and rd.dtype == rh.dtype and np.array_equal(rd, rh)
        for b in range(xs.shape[0]):
            for l in range(3):
                a, h = scipy.sparse.csc_matrix(cd[b][l]), scipy.sparse.csc_matrix(ch[b][l])
                assert a.shape == h.shape and (a != h).nnz == 0, (b, l)
                assert np.array_equal(a.indptr, h.indptr) and np.array_equal(a.indices, h.indices) and np.array_equal(a.data, h.data)
            assert ed[b].dtype == eh[b].dtype and np.array_equal(ed[b], eh[b]), b
            assert np.array_equal(ed[b], convertSparseMatricesToEvents(cd[b]))
        # residuals='energy': the same coefficients, and the energies of those residuals (summed on the device) instead of
        # their samples
        ce, en, _ = hcmp.computeCoefficientsBatch(xs, mld, memoryBudget=budget, epilogue='device', residuals='energy', **kw)
        assert en.shape == (xs.shape[0],) and en.dtype == np.float64
        assert np.allclose(en, np.sum(np.square(rd.reshape((xs.shape[0], -1))), axis=1), rtol=1e-12, atol=0.0)
        for b in range(xs.shape[0]):
            for l in range(3):
                assert (scipy.sparse.csc_matrix(ce[b][l]) != scipy.sparse.csc_matrix(cd[b][l])).nnz == 0, (b, l)
    hcmp.close()


@pytest.mark.gpu
def test_chunk_outside_the_epilogue_key_layout_is_finished_on_the_host(monkeypatch):
    """hscmp_hierarchy_epilogue answers HSCMP_ERR_UNSUPPORTED for shapes outside its sort keys (2^20 atoms or list entries,
    2^24 samples).  The default epilogue='device' then finishes that chunk on the host from the engines' slot lists instead of
    raising: same coefficient matrices, events, residual samples and energies as epilogue='host' (the refusal is injected
    here; any other error still surfaces)."""
    from hsc_amd import _native
    from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit
    z = _golden()
    mld = _mld().withSingletonBases()
    base = z['x']
    xs = np.stack([base, base[::-1].copy(), (base * np.float32(1.3)), np.roll(base, 37), np.roll(base, -11)])
    hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    kw = dict(toleranceSnr=[15.0, 20.0, 20.0], nbBlocks=4, singletonWeight=0.9, memoryBudget=2.5e6)
    ch, rh, _, eh = hcmp.computeCoefficientsBatch(xs, mld, epilogue='host', returnEvents=True, **kw)
    calls = []

    def refuse(self, *a, **k):
        calls.append(1)
        ex = _native.HscmpError('hscmp_hierarchy_epilogue failed (-5): shape outside the key layout')
        ex.code = _native.ERR_UNSUPPORTED
        raise ex

    with monkeypatch.context() as m:
        m.setattr(_native.Engine, 'hierarchy_epilogue', refuse)
        cd, rd, _, ed = hcmp.computeCoefficientsBatch(xs, mld, epilogue='device', returnEvents=True, **kw)
        ce, en, _ = hcmp.computeCoefficientsBatch(xs, mld, epilogue='device', residuals='energy', **kw)
    assert len(calls) >= 2                                     # (several chunks, each refused)
    assert rd.shape == rh.shape and np.array_equal(rd, rh)
    assert np.allclose(en, np.sum(np.square(rh.reshape((xs.shape[0], -1)).astype(np.float64)), axis=1), rtol=1e-12, atol=0.0)
    for b in range(xs.shape[0]):
        for l in range(3):
            a, h = scipy.sparse.csc_matrix(cd[b][l]), scipy.sparse.csc_matrix(ch[b][l])
            assert a.shape == h.shape and (a != h).nnz == 0, (b, l)
            assert (scipy.sparse.csc_matrix(ce[b][l]) != h).nnz == 0, (b, l)
        assert np.array_equal(ed[b], eh[b]), b

    def broken(self, *a, **k):
        ex = _native.HscmpError('hscmp_hierarchy_epilogue failed (-3): something else')
        ex.code = -3
        raise ex

    with monkeypatch.context() as m:
        m.setattr(_native.Engine, 'hierarchy_epilogue', broken)
        with pytest.raises(_native.HscmpError, match='something else'):
            hcmp.computeCoefficientsBatch(xs, mld, epilogue='device', **kw)
    hcmp.close()


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['config4_dims', 'three_levels_dense_rows'])
def test_level_input_buffer_is_clean_for_the_next_batch(case, monkeypatch):
    """Levels >= 1 read a dense [T, F] buffer that holds the previous level's coefficients.  Between batches the
    engine zeroes only the cells its row lists name (hscmp_api.hip: clear_listed_cells_kernel) instead of the whole
    buffer: batch A, then a DIFFERENT batch B through the same coder must equal B through a fresh coder that clears
    everything (HSCMP_NO_LAZY_CLEAR), bit for bit -- including inputs with rows that overflow their lists."""
    import hsc_amd.synth as synth
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    if case == 'config4_dims':
        mld = synth.make_hierarchy(W1=17, seed=4)
        mlds = mld.withSingletonBases()
        T, B = 8192, 24
        xa = synth.make_hierarchy_batch(mld, T, 0, B, seed=4)
        xb = synth.make_hierarchy_batch(mld, T, 1000, B, seed=9)
        kw = dict(toleranceSnr=[30.0, 30.0], nbBlocks=10, singletonWeight=0.9)
    else:
        z = _golden()
        mlds = _mld().withSingletonBases()
        rs = np.random.RandomState(3)
        base = z['x']
        xa = np.stack([base, base[::-1].copy(), np.roll(base, 11)]).astype(np.float32)
        # noise: level-0 codes with many atoms per position => level-1 rows with more cells than a row list holds
        xb = (0.5 * rs.standard_normal(xa.shape)).astype(np.float32)
        kw = dict(toleranceSnr=[12.0, 14.0, 14.0], nbBlocks=4, singletonWeight=0.9)
    warm = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    warm.computeCoefficientsBatch(xa, mlds, **kw)
    cb, rb, _ = warm.computeCoefficientsBatch(xb, mlds, **kw)
    ca, ra, _ = warm.computeCoefficientsBatch(xa[:2], mlds, **kw)          # and a smaller batch after a larger one
    monkeypatch.setenv('HSCMP_NO_LAZY_CLEAR', '1')
    fresh = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    cb2, rb2, _ = fresh.computeCoefficientsBatch(xb, mlds, **kw)
    ca2, ra2, _ = fresh.computeCoefficientsBatch(xa[:2], mlds, **kw)
    assert np.array_equal(rb, rb2) and np.array_equal(ra, ra2)
    for got, exp in ((cb, cb2), (ca, ca2)):
        for b in range(len(exp)):
            for l in range(len(exp[b])):
                assert (scipy.sparse.csc_matrix(got[b][l]) != scipy.sparse.csc_matrix(exp[b][l])).nnz == 0, (b, l)
    warm.close(); fresh.close()


# ---- the level pipeline object (hierarchical._LevelPipeline), step by step on the CPU ------------------------------------------

class _FakeEngine(object):
    def __init__(self, total=100e9):
        self.total = total
        self.copied = []

    def mem_info(self):
        return (self.total, self.total)

    def copy_from_device(self, ptr, shape, dtype):
        self.copied.append((int(ptr), tuple(shape), np.dtype(dtype)))
        return np.full(shape, 7, dtype=dtype)


def _pipeline(xs, **over):
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit, _LevelPipeline
    mld = _mld().withSingletonBases()
    kw = dict(toleranceSnr=[10.0, 20.0, 30.0], nbBlocks=4, singletonWeight=0.25, returnDistributed=True, epilogue='device', returnEvents=False,
              deviceInput=None, residuals='samples', memoryBudget=None)
    kw.update(over)
    return _LevelPipeline(HierarchicalConvolutionalMatchingPursuit(method='cmp'), xs, mld, kw['toleranceSnr'], kw['nbBlocks'], kw['singletonWeight'],
                          kw['returnDistributed'], kw['epilogue'], kw['returnEvents'], kw['deviceInput'], kw['residuals'], kw['memoryBudget']), mld


def test_pipeline_level_setup_weights_and_targets():
    """:1439-1450: per-level SNR target (list or scalar), singletonWeight on the first K_l - countsNoSingletons[l] atoms, eps of the
    dictionary's dtype."""
    xs = np.zeros((3, 512), dtype=np.float32)
    pipe, mld = _pipeline(xs)
    for level in range(3):
        D, w, snr, eps = pipe.level_setup(level)
        ns = D.shape[0] - mld.countsNoSingletons[level]
        assert snr == [10.0, 20.0, 30.0][level] and eps == float(np.finfo(D.dtype).eps)
        assert np.all(w[:ns] == 0.25) and np.all(w[ns:] == 1.0) and w.dtype == D.dtype
    pipe, _ = _pipeline(xs, toleranceSnr=12.5)
    assert [pipe.level_setup(l)[2] for l in range(3)] == [12.5] * 3


def test_pipeline_chunk_size_follows_the_memory_budget():
    from hsc_amd import _native
    xs = np.zeros((40, 512), dtype=np.float32)
    pipe, mld = _pipeline(xs)
    pipe.engines = [_FakeEngine(total=10e6)]
    setups = [None] + [pipe.level_setup(l) for l in (1, 2)]
    stats0 = np.zeros((40, _native.STAT_COUNT), dtype=np.int32)
    stats0[:, _native.STAT_SLOTS] = 50
    per_signal = sum(1.05 * 512 * setups[l][0].shape[2] * 8 + 160 * 512 + 80.0 * 4096 for l in (1, 2))
    assert pipe.chunk_size(setups, stats0) == int(0.6 * 10e6 // per_signal)        # default: 60 % of the GPU
    pipe.memoryBudget = 3.5 * per_signal
    assert pipe.chunk_size(setups, stats0) == 3
    pipe.memoryBudget = 1.0
    assert pipe.chunk_size(setups, stats0) == 1                                     # never below one signal
    pipe.memoryBudget = 1e15
    assert pipe.chunk_size(setups, stats0) == 40                                    # never beyond the batch
    stats0[:, _native.STAT_SLOTS] = 100000                                          # long slot lists count too
    pipe.memoryBudget = 3.5 * per_signal
    assert pipe.chunk_size(setups, stats0) < 3


def test_pipeline_host_signal_reads_the_device_when_the_batch_came_as_a_pointer():
    """with deviceInput the host array is a placeholder (shape and dtype only); whatever falls back to the host must copy the
    signal from the device."""
    xs = np.zeros((5, 512), dtype=np.float32)
    pipe, _ = _pipeline(xs, deviceInput=4096)
    eng = _FakeEngine()
    pipe.engines, pipe.dt0 = [eng], np.float32
    got = pipe.host_signal(3)
    assert eng.copied == [(4096 + 3 * 512 * 4, (512,), np.dtype(np.float32))] and np.all(got == 7)
    pipe, _ = _pipeline(np.arange(10, dtype=np.float32).reshape((2, 5)))
    assert np.array_equal(pipe.host_signal(1), [5, 6, 7, 8, 9])                     # host batch: the row itself


def test_pipeline_capacity_hint_and_regrowth_bound():
    from hsc_amd import _native
    xs = np.zeros((2, 512), dtype=np.float32)
    pipe, _ = _pipeline(xs)
    eng = _FakeEngine()
    assert pipe.start_capacity(eng, 4096) == 4096
    eng._capacity_hint = ((512, 4), 20000)
    assert pipe.start_capacity(eng, 4096) == min(20000, _native.max_event_capacity(512))
    eng._capacity_hint = ((1024, 4), 20000)                                        # another shape: no hint
    assert pipe.start_capacity(eng, 4096) == 4096
