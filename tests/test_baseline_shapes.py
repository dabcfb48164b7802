"""The BASELINE.json shapes themselves under asserts.

  config 2  B=1024 signals of 65536 samples, 256 x 64 dictionary, L0=256 -- the workload bench.py times: two rounds of
            512 co-resident workgroups (or one round of the four-signals-per-workgroup loop), never reached by the
            small-batch parity tests.  The 8 signals the REAL reference encoded (tests/golden/cmp_config.npz) sit at spread
            batch positions, two more positions are checked bit for bit against the CPU oracle, every signal is checked
            through size-independent properties, and a second run must reproduce the first bit for bit.
  config 4  2-level hierarchy at the real dictionary dimensions (256 x 64, then (256+128) x 16 x 256) with more than two
            signals per CU, so that the four-workgroups-per-CU build of the level loop is the one dispatched; 8192 samples
            so that the CPU oracle finishes.  Signal 0 against the REAL reference (tests/golden/hsc_config4.npz), a few
            against the host logic on the oracle level coder, all of them against the per-signal path / properties.

CPU part: the oracle (as level coder of the host logic) against the same reference goldens, and reconstructSignal's
dense branch (modeling.py:247-258) against the reference's fftconvolve result."""
import os

import numpy as np
import pytest
import scipy.sparse

import golden_util as gu

SPREAD = [0, 255, 256, 511, 512, 700, 767, 1023]          # batch positions of the 8 golden signals
GOLDEN8 = [('planted', 0), ('planted', 1), ('planted', 2), ('planted', 3), ('noise', 0), ('noise', 1), ('noise', 2), ('noise', 3)]
HSC_KW = dict(toleranceSnr=[30.0, 40.0], nbBlocks=10, singletonWeight=0.95)


class _OracleLevelCoder(object):
    def __init__(self, D):
        self.D = D

    def encode(self, X, **kw):
        from oracle import hsc_oracle as orc
        coefficients, residual, _ = orc.cmp_encode(np.asarray(X), self.D, **kw)
        return coefficients, residual


def _hierarchy(W1):
    import hsc_amd.synth as synth
    mld = synth.make_hierarchy(W1=W1, seed=4)
    return mld, mld.withSingletonBases()


def _check_config4_golden(coefficients, residual, x, W1):
    z = gu.load('hsc_config4.npz')
    for l, c in enumerate(coefficients):
        row, col, data = gu.csc_triplets(scipy.sparse.csc_matrix(c))
        assert np.array_equal(row, z['w%d__level%d_row' % (W1, l)]), 'level %d positions' % l
        assert np.array_equal(col, z['w%d__level%d_col' % (W1, l)]), 'level %d atoms' % l
        assert gu.rel_err(data, z['w%d__level%d_data' % (W1, l)]) <= 1e-5
    e = float(np.sum(np.square(np.asarray(residual, dtype=np.float64))))
    assert abs(e - float(z['w%d__residual_energy' % W1])) <= 1e-5 * float(z['w%d__residual_energy' % W1])


@pytest.mark.parametrize('W1', [16, 17])
def test_config4_dims_oracle_level_coder_vs_reference_golden(W1, monkeypatch):
    """Pins the oracle at the config-4 dictionary dimensions: host logic + oracle level coder == the real reference."""
    import hsc_amd.synth as synth
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    z = gu.load('hsc_config4.npz')
    mld, mlds = _hierarchy(W1)
    assert synth.digest(*mlds.dictionaries) == str(z['w%d__dict_digest' % W1])
    x = synth.make_hierarchy_signal(mld, 8192, 0, seed=4)
    assert synth.digest(x) == str(z['w%d__x_digest' % W1])
    ref = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    monkeypatch.setattr(ref, '_level_coder', lambda D: _OracleLevelCoder(D))
    coefficients, residual = ref.computeCoefficients(x, mlds, **HSC_KW)
    _check_config4_golden(coefficients, residual, x, W1)
    snr = 10 * np.log10(np.sum(x.astype(np.float64) ** 2) / np.sum(np.square(residual)))
    # the 16-tap shape of BASELINE config 4 reconstructs every level-1 pattern one sample late (synth.make_hierarchy)
    assert (snr > 29.0) if W1 == 17 else (snr < 5.0)


@pytest.mark.parametrize('name', ['dense1d', 'dense1d_odd', 'dense2d'])
def test_reconstruct_signal_dense_branch_vs_reference(name):
    """modeling.py:247-258 synthesises a DENSE coefficient array with fftconvolve; the engine's overlap-add must agree
    (to fft rounding), and with the reference's own sparse branch."""
    from hsc_amd.modeling import reconstructSignal
    z = gu.load('hsc_config4.npz')
    D, C = z[name + '__D'], z[name + '__C']
    got = reconstructSignal(C, D)
    exp = z[name + '__signal']
    assert got.shape == exp.shape and got.dtype == exp.dtype
    assert float(np.max(np.abs(got - exp))) <= 1e-10 * max(1.0, float(np.max(np.abs(exp))))
    got_s = reconstructSignal(scipy.sparse.csc_matrix(C), D)
    assert float(np.max(np.abs(got_s - z[name + '__signal_sparse']))) <= 1e-12


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_config2_full_batch_1024():
    import hsc_amd.synth as synth
    from hsc_amd import _native
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    z = gu.load('cmp_config.npz')
    B, T, L0 = 1024, 65536, 256
    D = synth.make_dictionary(256, 64, seed=2)
    xs = synth.make_batch(D, T, 8, B, kind='planted', nb_atoms=L0, seed=2)      # fill: planted signals 8 .. 1031
    for pos, (kind, idx) in zip(SPREAD, GOLDEN8):
        xs[pos] = synth.make_signal(D, T, idx, kind=kind, nb_atoms=L0, seed=2)
        assert synth.digest(xs[pos]) == str(z['config2_%s_%d__x_digest' % (kind, idx)])
    cmp = ConvolutionalMatchingPursuit()
    res = cmp.computeCoefficientsBatch(xs, D, nbNonzeroCoefs=L0)
    assert res.variant.startswith('mfma_init+mfma_loop')
    # (i) the 8 signals of the real reference: indices exact, coefficients / residual energy 1e-5
    for pos, (kind, idx) in zip(SPREAD, GOLDEN8):
        name = 'config2_%s_%d' % (kind, idx)
        t, k, c = res.events[pos]
        assert np.array_equal(t, z[name + '__t']) and np.array_equal(k, z[name + '__k']), (pos, name)
        assert gu.rel_err(c, z[name + '__c']) <= 1e-5
        row, col, data = gu.csc_triplets(res.coefficients[pos])
        assert np.array_equal(row, z[name + '__csc_row']) and np.array_equal(col, z[name + '__csc_col'])
        assert gu.rel_err(data, z[name + '__csc_data']) <= 1e-5
        e = float(np.sum(np.square(res.residuals[pos].astype(np.float64))))
        assert abs(e - float(z[name + '__residual_energy'])) <= 1e-5 * float(z[name + '__residual_energy'])
    # (ii) one position of each half of the batch against the CPU oracle, bit for bit
    for pos in (300, 900):
        coef, r, info = orc.cmp_encode(xs[pos], D, nbNonzeroCoefs=L0)
        t, k, c = res.events[pos]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), pos
        assert np.array_equal(res.residuals[pos], r), pos
    # (iii) every signal: stop rule, counts, tracked residual energy == recomputed, events consistent with the slots
    st = res.stats
    assert np.all(st[:, _native.STAT_STOP] == 2), np.bincount(st[:, _native.STAT_STOP])       # HSCMP_STOP_NNZ
    assert np.all(st[:, _native.STAT_NNZ] == L0)
    assert np.all(st[:, _native.STAT_EVENTS] == st[:, _native.STAT_ITERATIONS])
    assert np.all(st[:, _native.STAT_ITERATIONS] == st[:, _native.STAT_NNZ] + st[:, _native.STAT_DUPLICATES])
    e_rec = np.sum(np.square(res.residuals.astype(np.float64)), axis=1)
    assert np.all(np.abs(e_rec - res.energies[:, 1]) <= 1e-5 * res.energies[:, 0])      # drift of the running f32 difference
    assert np.all(res.energies[:, 1] < res.energies[:, 0])
    for b in range(B):
        t, k, c = res.events[b]
        acc = scipy.sparse.coo_matrix((c.astype(np.float64), (t, k)), shape=(T, 256)).tocsc()
        assert abs(acc - res.coefficients[b]).max() <= 1e-12
        assert L0 - 2 <= res.coefficients[b].nnz <= L0      # (an accumulated coefficient may cancel to exactly zero)
    # (iv) determinism at batch scale: a second run reproduces events, slots and residuals bit for bit -- once with the
    # dispatcher's choice (four signals per workgroup at this batch size), once with one signal per workgroup
    assert res.variant.endswith('_x4')
    for force in (None, '0'):
        if force is not None:
            os.environ['HSCMP_MFMA_QUAD'] = force
        try:
            res2 = ConvolutionalMatchingPursuit().computeCoefficientsBatch(xs, D, nbNonzeroCoefs=L0)
        finally:
            os.environ.pop('HSCMP_MFMA_QUAD', None)
        assert res2.variant.endswith('_x4') == (force is None)
        _same_results(res, res2, B)


def _same_results(res, res2, B):
    assert np.array_equal(res.stats, res2.stats) and np.array_equal(res.energies, res2.energies)
    assert np.array_equal(res.residuals, res2.residuals)
    for b in range(B):
        assert all(np.array_equal(u, v) for u, v in zip(res.events[b], res2.events[b])), b
        assert (res.coefficients[b] != res2.coefficients[b]).nnz == 0, b


@pytest.mark.gpu
def test_config4_real_dictionary_dims_packed_batch(monkeypatch):
    """Exact config-4 dictionary shape (16 taps at level 1), 640 signals of 8192 samples.  The dispatcher runs both levels on
    the round-parallel loops (csrc/hscmp_rp*.h); with HSCMP_RP=0 it falls to the loops of round 2 -- more than two signals
    per CU: four signals per workgroup at level 0, the four-workgroups-per-CU build at level 1 -- and all of them must agree
    on every signal, bit for bit."""
    import hsc_amd.synth as synth
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    B, T, W1 = 640, 8192, 16
    mld, mlds = _hierarchy(W1)
    xs = synth.make_hierarchy_batch(mld, T, 0, B, seed=4)
    gpu = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    coefs, residuals, timings = gpu.computeCoefficientsBatch(xs, mlds, **HSC_KW)
    assert timings[0]['variant'].startswith('mfma_init+mfma_loop') and timings[1]['variant'].startswith('dictlist_init+dictlist_loop')
    assert timings[1].get('chunks', 1) == 1
    # signal 0: the REAL reference
    _check_config4_golden(coefs[0], residuals[0], xs[0], W1)
    # a few signals: host logic on the CPU oracle as level coder, bit for bit
    ref = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    monkeypatch.setattr(ref, '_level_coder', lambda D: _OracleLevelCoder(D))
    for b in (1, 333, 639):
        exp_c, exp_r = ref.computeCoefficients(xs[b], mlds, **HSC_KW)
        for l in range(2):
            assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(exp_c[l])).nnz == 0, (b, l)
        assert np.array_equal(residuals[b], exp_r), b
    # spread signals: the batch (packed build) equals the per-signal path (roomy build), bit for bit
    for b in (64, 257, 513, 600):
        c1, r1 = gpu.computeCoefficients(xs[b], mlds, **HSC_KW)
        for l in range(2):
            assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(c1[l])).nnz == 0, (b, l)
        assert np.array_equal(residuals[b], r1), b
    # every signal: the residual is x minus the synthesis of the returned coefficients (independent numpy overlap-add)
    reps = mlds.getMultiscaleDictionaries()
    for b in range(0, B, 7):
        recon = np.zeros(T)
        for l in range(2):
            m = scipy.sparse.coo_matrix(coefs[b][l])
            sc = reps[l].shape[1]
            for t, k, c in zip(m.row, m.col, m.data):
                lo = t - (sc - 1) // 2
                s, e = max(lo, 0), min(lo + sc, T)
                recon[s:e] += c * reps[l][k][s - lo:e - lo].astype(np.float64)
        assert float(np.max(np.abs((xs[b] - recon) - residuals[b]))) <= 1e-9, b
    assert all(c[0].shape == (T, 256) and c[1].shape == (T, 384) for c in coefs)
    # determinism at batch scale: the round-parallel loops (dispatched here) against the four-signals-per-workgroup level-0
    # loop and the packed level-1 loop (HSCMP_RP=0: B > 2 x CUs) and against the one-signal-per-workgroup loop, on ALL
    # signals, then three more runs of the dispatched build.  (This kind of comparison is what exposed the reflected-sample
    # load/store race of the fused atom body: about one signal in a thousand)
    assert timings[0]['variant'].endswith('_rp') and timings[1]['variant'].endswith('_rp')
    monkeypatch.setenv('HSCMP_RP', '0')
    coefsq, residualsq, tq = gpu.computeCoefficientsBatch(xs, mlds, **HSC_KW)
    assert tq[0]['variant'].endswith('_x4') and not tq[1]['variant'].endswith('_rp')
    monkeypatch.setenv('HSCMP_MFMA_QUAD', '0')
    coefs1, residuals1, t1 = gpu.computeCoefficientsBatch(xs, mlds, **HSC_KW)
    assert not t1[0]['variant'].endswith('_x4') and not t1[0]['variant'].endswith('_rp')
    monkeypatch.delenv('HSCMP_MFMA_QUAD')
    monkeypatch.delenv('HSCMP_RP')
    assert np.array_equal(residuals, residuals1) and np.array_equal(residuals, residualsq)
    for b in range(B):
        for l in range(2):
            assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(coefsq[b][l])).nnz == 0, (b, l)
    for rep in range(3):
        coefs2, residuals2, _ = gpu.computeCoefficientsBatch(xs, mlds, **HSC_KW)
        assert np.array_equal(residuals, residuals2), rep
        for b in range(B):
            for l in range(2):
                assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(coefs2[b][l])).nnz == 0, (rep, b, l)
                assert rep or (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(coefs1[b][l])).nnz == 0, (b, l)
    gpu.close()


@pytest.mark.gpu
def test_config4_consistent_hierarchy_reconstructs():
    """17 taps at level 1 (scales [64, 80]: the nearest shape whose hierarchy is self-consistent): the 2-level encode
    reproduces its input to the level-0 tolerance, and signal 0 equals the REAL reference's encode."""
    import hsc_amd.synth as synth
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    B, T, W1 = 48, 8192, 17
    mld, mlds = _hierarchy(W1)
    xs = synth.make_hierarchy_batch(mld, T, 0, B, seed=4)
    gpu = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    coefs, residuals, timings = gpu.computeCoefficientsBatch(xs, mlds, **HSC_KW)
    _check_config4_golden(coefs[0], residuals[0], xs[0], W1)
    snr = 10 * np.log10(np.sum(xs.astype(np.float64) ** 2, axis=1) / np.sum(np.square(residuals), axis=1))
    assert np.all(snr >= 25.0), snr.min()
    assert sum(c[1][:, 256:].nnz for c in coefs) > 10 * B        # composite atoms are actually used
    gpu.close()


@pytest.mark.gpu
def test_config5_generated_three_level_dictionary(monkeypatch):
    """BASELINE configs[4] at its dictionary dimensions: generated Perlin dictionary, scales [32, 64, 128] (taps 32 / 33 /
    65), 4x overcomplete per level plus singleton bases => level dictionaries 128x32, (128+132)x33x128, (260+260)x65x260;
    Poisson-event signals at compression 0.25 (bench_hsc.build_workload).  24 signals of 8192 samples through the
    device-chained batch path: two of them against the host logic with the CPU oracle as level coder, bit for bit on all
    three levels; all of them against the per-signal path property (batch == single) on a sample, the independent
    reconstruction check, and a second run."""
    import bench_hsc
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    B, T = 24, 8192
    mlds, xs, kw, _ = bench_hsc.build_workload(5, B, T, 0)
    assert [tuple(mlds.getRawDictionary(l).shape) for l in range(3)] == [(128, 32), (260, 33, 128), (520, 65, 260)]
    gpu = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    coefs, residuals, timings = gpu.computeCoefficientsBatch(xs, mlds, **kw)
    assert timings[0]['variant'].startswith('mfma_init+mfma_loop') and all(t['variant'].startswith('dictlist_init+dictlist_loop') for t in timings[1:])
    ref = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    monkeypatch.setattr(ref, '_level_coder', lambda D: _OracleLevelCoder(D))
    for b in (0, 17):
        exp_c, exp_r = ref.computeCoefficients(xs[b], mlds, **kw)
        for l in range(3):
            assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(exp_c[l])).nnz == 0, (b, l)
        assert np.array_equal(residuals[b], exp_r), b
    for b in (5, 23):
        c1, r1 = gpu.computeCoefficients(xs[b], mlds, **kw)
        for l in range(3):
            assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(c1[l])).nnz == 0, (b, l)
        assert np.array_equal(residuals[b], r1), b
    reps = mlds.getMultiscaleDictionaries()
    snr = []
    for b in range(B):
        recon = np.zeros(T)
        for l in range(3):
            m = scipy.sparse.coo_matrix(coefs[b][l])
            sc = reps[l].shape[1]
            for t, k, c in zip(m.row, m.col, m.data):
                lo = t - (sc - 1) // 2
                s, e = max(lo, 0), min(lo + sc, T)
                recon[s:e] += c * reps[l][k][s - lo:e - lo].astype(np.float64)
        assert float(np.max(np.abs((xs[b] - recon) - residuals[b]))) <= 1e-9, b
        snr.append(10 * np.log10(np.sum(xs[b].astype(np.float64) ** 2) / np.sum(residuals[b] ** 2)))
    assert min(snr) >= 20.0, min(snr)
    coefs2, residuals2, _ = gpu.computeCoefficientsBatch(xs, mlds, **kw)
    assert np.array_equal(residuals, residuals2)
    for b in range(B):
        for l in range(3):
            assert (scipy.sparse.csc_matrix(coefs[b][l]) != scipy.sparse.csc_matrix(coefs2[b][l])).nnz == 0, (b, l)
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize('config,B,pack', [(5, 6, None), (4, 12, None), (4, 16, '4')])
def test_locomp_at_baseline_shapes_side_by_side_equals_sequential(config, B, pack, monkeypatch):
    """The reference's default method on BASELINE configs[4] / [3] at their FULL signal length (T = 65536, ten blocks): everything the
    device loop does beside the selection-by-selection order -- groups computed ahead one wave each (up to 54 atoms on the sparse levels,
    their Gram matrices in the waves' row-pipeline slots), committed by their waves alone, rows re-correlated a batch at a time
    (HSCMP_LOCOMP_AHEAD=7, the default) -- against HSCMP_LOCOMP_AHEAD=0, level by level on the same inputs: stop reasons, counters, event
    records, coefficients and residuals bit for bit.  The third level of config 5 is where groups of more than 32 atoms occur; the third case
    runs config 4 with FOUR signals per workgroup on its first level (HSCMP_LOCOMP_PACK=4, what the full batch of 1024 takes: each signal's
    four waves meet at a counter in LDS instead of the hardware barrier and a wave can be a whole phase late -- the arrangement under which a
    selection's verdict must survive the next one's)."""
    import bench_hsc
    from hsc_amd.modeling import LoCOMP
    if pack:
        monkeypatch.setenv('HSCMP_LOCOMP_PACK', pack)
    mlds, xs, kw, _ = bench_hsc.build_workload(config, B, 65536, 0, 17)
    inp = xs
    for level in range(mlds.getNbLevels()):
        D = mlds.getRawDictionary(level)
        nbS = D.shape[0] - mlds.countsNoSingletons[level]
        w = np.ones((D.shape[0],), dtype=D.dtype); w[:nbS] = kw.get('singletonWeight', 0.5)
        args = dict(toleranceSnr=kw['toleranceSnr'][level], nbBlocks=kw['nbBlocks'], weights=w)
        monkeypatch.setenv('HSCMP_LOCOMP_AHEAD', '7')
        fast = LoCOMP().computeCoefficientsBatch(inp, D, **args)
        monkeypatch.setenv('HSCMP_LOCOMP_AHEAD', '0')
        seq = LoCOMP().computeCoefficientsBatch(inp, D, **args)
        if pack and level == 0:
            assert 'mfma' in fast.variant, fast.variant
        assert np.array_equal(fast.stats, seq.stats), level
        assert np.array_equal(fast.residuals, seq.residuals), level
        for b in range(B):
            assert (fast.coefficients[b] != seq.coefficients[b]).nnz == 0, (level, b)
            assert all(np.array_equal(u, v) for u, v in zip(fast.events[b], seq.events[b])), (level, b)
        assert int(fast.stats[:, 4].min()) > 100, level                  # (hundreds of selections per signal)
        inp = np.stack([c.toarray() for c in fast.coefficients], axis=0)


@pytest.mark.gpu
@pytest.mark.parametrize('method', ['cmp', 'locomp'])
def test_config4_dims_several_chunks_through_the_device_epilogue(method):
    """The memory-budget path that anything larger than config 4 takes by default: levels >= 1 walk the batch in chunks (their dense
    float64 input [T, 256] lives on the device: 16.8 MB per signal here, 134 MB at T = 65536), every chunk through the device epilogue.
    A budget that holds 7 of the 20 signals => three chunks (7 + 7 + 6); the result must equal the one-chunk run bit for bit --
    coefficient matrices, event records, residual samples and device-summed residual energies -- for both methods."""
    import hsc_amd.synth as synth
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    B, T, W1 = 20, 8192, 17
    mld, mlds = _hierarchy(W1)
    xs = synth.make_hierarchy_batch(mld, T, 0, B, seed=4)
    h = HierarchicalConvolutionalMatchingPursuit(method=method)
    try:
        one = h.computeCoefficientsBatch(xs, mlds, returnEvents=True, **HSC_KW)
        assert one[2][1]['chunks'] == 1
        # the budget of 7 signals, by the pipeline's own estimate (_LevelPipeline.chunk_size; tests/test_hierarchical.py pins it)
        per_signal = 1.05 * T * 256 * 8 + 160 * T + 80.0 * 4096
        budget = 7.5 * per_signal
        many = h.computeCoefficientsBatch(xs, mlds, returnEvents=True, memoryBudget=budget, **HSC_KW)
        assert many[2][1]['chunks'] == 3, many[2][1]
        energy = h.computeCoefficientsBatch(xs, mlds, residuals='energy', memoryBudget=budget, **HSC_KW)
        assert energy[2][1]['chunks'] == 3
    finally:
        h.close()
    assert np.array_equal(one[1], many[1])
    assert np.allclose(energy[1], np.sum(np.square(one[1]), axis=1), rtol=1e-12, atol=0.0)
    for b in range(B):
        assert np.array_equal(one[3][b], many[3][b]), b
        for l in range(2):
            assert (scipy.sparse.csc_matrix(one[0][b][l]) != scipy.sparse.csc_matrix(many[0][b][l])).nnz == 0, (b, l)
            assert (scipy.sparse.csc_matrix(one[0][b][l]) != scipy.sparse.csc_matrix(energy[0][b][l])).nnz == 0, (b, l)


@pytest.mark.gpu
def test_config4_full_length_signals_level0_vs_oracle_level1_vs_dense_input():
    """BASELINE configs[3] at its FULL signal length (T = 65536, 17 taps), 32 signals.  The oracle cannot run the dense level-1
    correlation at this size (219 GFLOP per signal), so the levels are pinned separately: level 0 of two signals against the CPU oracle,
    bit for bit (positions, atoms, coefficients, residual); level 1 of the same two signals -- device-chained: the input scattered from
    the level-0 slots in GPU memory -- against the single-level engine run on the DENSE [T, 256] float64 input the host builds from the
    level-0 matrix (the path the T = 8192 oracle tests pin), bit for bit; every signal: reconstruction SNR and the device energies."""
    import hsc_amd.synth as synth
    from oracle import hsc_oracle as orc
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    B, T, W1 = 32, 65536, 17
    mld, mlds = _hierarchy(W1)
    xs = synth.make_hierarchy_batch(mld, T, 0, B, seed=4)
    h = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    try:
        raw, residuals, timings = h.computeCoefficientsBatch(xs, mlds, returnDistributed=False, epilogue='host', **HSC_KW)     # (level matrices as encoded)
        coefs, energies, _ = h.computeCoefficientsBatch(xs, mlds, residuals='energy', **HSC_KW)
    finally:
        h.close()
    D0, D1 = mlds.getRawDictionary(0), mlds.getRawDictionary(1)
    w1 = np.ones(D1.shape[0], dtype=D1.dtype); w1[:D1.shape[0] - mlds.countsNoSingletons[1]] = HSC_KW['singletonWeight']
    first = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    try:
        lvl0, _, _ = first.computeCoefficientsBatch(xs[[3, 29]], _first_level_only(mlds), toleranceSnr=[HSC_KW['toleranceSnr'][0]], nbBlocks=HSC_KW['nbBlocks'],
                                                    singletonWeight=HSC_KW['singletonWeight'], returnDistributed=False)
    finally:
        first.close()
    for j, b in enumerate((3, 29)):
        c0, r0, info = orc.cmp_encode(xs[b], D0, toleranceSnr=HSC_KW['toleranceSnr'][0], nbBlocks=HSC_KW['nbBlocks'],
                                      weights=np.ones(D0.shape[0], dtype=D0.dtype))
        assert (scipy.sparse.csc_matrix(lvl0[j][0]) != c0).nnz == 0, b
        dense = np.asarray(c0.toarray(), dtype=np.float64)
        res1 = ConvolutionalMatchingPursuit().computeCoefficientsBatch(dense[None], D1, toleranceSnr=HSC_KW['toleranceSnr'][1], nbBlocks=HSC_KW['nbBlocks'], weights=w1)
        assert (scipy.sparse.csc_matrix(raw[b][1]) != res1.coefficients[0]).nnz == 0, b
    snr = 10 * np.log10(np.sum(xs.astype(np.float64) ** 2, axis=1) / energies)
    assert np.all(snr >= 25.0), snr.min()
    # (the non-distributed host-epilogue run sums the same atoms from the last level's matrix alone: the same residual up to summation order)
    assert np.allclose(energies, np.sum(np.square(residuals), axis=1), rtol=1e-6, atol=0.0)


def _first_level_only(mlds):
    from hsc_amd.dataset import MultilevelDictionary
    return MultilevelDictionary.fromRawDictionaries([mlds.getRawDictionary(0)], [int(mlds.scales[0])])
