"""Dataset synthesis (reference hsc/dataset.py:412-796): Perlin base atoms,
random decompositions, Poisson event streams, rendered signals.  The generators draw from numpy's
RandomState in the reference's order, so under the seeds of tests/golden/synth_small.npz (written by
tools/make_golden.py from the real reference) everything must be reproduced bit for bit."""
import numpy as np
import pytest

import golden_util as gu


def _generate(name, rng_instance=False):
    from hsc_amd.dataset import MultilevelDictionaryGenerator
    kw, dseed, rate, eseed, n = gu.SYNTH_CASES[name]
    if rng_instance:
        return MultilevelDictionaryGenerator(np.random.RandomState(dseed)).generate(**kw)
    np.random.seed(dseed)
    return MultilevelDictionaryGenerator().generate(**kw)


@pytest.mark.parametrize('rng_instance', [False, True])
@pytest.mark.parametrize('name', sorted(gu.SYNTH_CASES))
def test_generated_dictionary_matches_reference(name, rng_instance):
    z = gu.load('synth_small.npz')
    mld = _generate(name, rng_instance)
    assert mld.getNbLevels() == int(z[name + '__nlevels'])
    assert int(mld.hasSingletonBases) == int(z[name + '__singletons'])
    for l in range(mld.getNbLevels()):
        for key, got in (('raw', mld.dictionaries[l]), ('rep', mld.representations[l])):
            exp = z['%s__%s%d' % (name, key, l)]
            assert got.dtype == exp.dtype and got.shape == exp.shape and np.array_equal(got, exp), (key, l)
        if l > 0:
            for j, entry in enumerate(mld.decompositions[l - 1]):
                for q, part in enumerate(entry):
                    assert np.array_equal(np.asarray(part), z['%s__dec%d_%d_%d' % (name, l, j, q)])
    # unit-norm, localised atoms
    for rep in mld.representations:
        assert np.allclose(np.sqrt(np.sum(np.square(rep.astype(np.float64)), axis=1)), 1.0, atol=1e-5)


@pytest.mark.parametrize('tag,ratio', [('plain', None), ('scaled', 0.25)])
@pytest.mark.parametrize('name', sorted(gu.SYNTH_CASES))
def test_generated_events_and_signal_match_reference(name, tag, ratio):
    from hsc_amd.dataset import SignalGenerator, EVENT_DTYPE
    z = gu.load('synth_small.npz')
    kw, dseed, rate, eseed, n = gu.SYNTH_CASES[name]
    mld = _generate(name)
    np.random.seed(eseed)
    gen = SignalGenerator(mld, [rate] * mld.getNbLevels())
    res = gen.generateEvents(n, ratio)
    events = res if ratio is None else res[0]
    if ratio is not None:
        assert np.array_equal(np.asarray(res[1], dtype=np.float64), z['%s__%s_rates' % (name, tag)])
    assert events.dtype == EVENT_DTYPE
    for f in events.dtype.names:
        assert np.array_equal(events[f], z['%s__%s_ev_%s' % (name, tag, f)]), f
    signal = gen.generateSignalFromEvents(events, nbSamples=n)
    exp = z['%s__%s_signal' % (name, tag)]
    assert signal.dtype == exp.dtype and np.array_equal(signal, exp)
    assert len(gen.generateSignalFromEvents(events)) == int(z['%s__%s_autolen' % (name, tag)])
    # events respect the borders of their pattern (hsc/dataset.py:735-741)
    for t, l in zip(events['f0'], events['f1']):
        sc = int(mld.scales[l])
        assert t >= (sc // 2 - 1 if sc % 2 == 0 else sc // 2) and t <= n - sc // 2


def test_rate_scaling_respects_the_bit_budget():
    from hsc_amd.dataset import SignalGenerator
    from hsc_amd.analysis import calculateBitForDatatype, calculateMultilevelInformationRates
    mld = _generate('no_overlap_nonneg')
    gen = SignalGenerator(mld, [0.05, 0.05], rng=np.random.RandomState(1))
    events, rates = gen.generateEvents(3000, 0.25)
    assert np.max(rates) < 0.05                                  # the initial rates were too high
    info = calculateMultilevelInformationRates(mld, np.copy(rates), 3000, dtype=np.float32)
    assert info[0] <= 0.25 * calculateBitForDatatype(np.float32) and len(info) == 2 and info[1] <= info[0]
    assert calculateBitForDatatype(np.float32) == 32 and calculateBitForDatatype(np.float64) == 64
    assert calculateBitForDatatype(np.int16) == 16


def test_live_reference_agrees_when_present():
    """In the build container the same comparison runs against the reference itself (other seeds)."""
    from oracle import ref_loader
    ref = ref_loader.load_reference()
    if ref is None:
        pytest.skip('reference not available here')
    from hsc_amd.dataset import MultilevelDictionaryGenerator, SignalGenerator
    kw = dict(scales=[20, 50], counts=[7, 5], decompositionSize=2, multilevelDecomposition=False, maxNbPatternsConsecutiveRejected=15)
    np.random.seed(21); a = ref.dataset.MultilevelDictionaryGenerator().generate(**kw)
    np.random.seed(21); b = MultilevelDictionaryGenerator().generate(**kw)
    assert all(np.array_equal(x, y) for x, y in zip(a.dictionaries, b.dictionaries))
    assert all(np.array_equal(x, y) for x, y in zip(a.representations, b.representations))
    np.random.seed(22); ea = ref.dataset.SignalGenerator(a, [0.003, 0.003]).generateEvents(3000)
    np.random.seed(22); eb = SignalGenerator(b, [0.003, 0.003]).generateEvents(3000)
    assert all(np.array_equal(ea[f], eb[f]) for f in ea.dtype.names)


def test_up_to_level_and_pickle_round_trip(tmp_path):
    from hsc_amd.dataset import MultilevelDictionary
    mld = _generate('cross_level')
    low = mld.upToLevel(1)
    assert low.getNbLevels() == 2 and np.array_equal(low.representations[1], mld.representations[1])
    base = mld.upToLevel(0)
    assert base.getNbLevels() == 1 and np.array_equal(base.getBaseDictionary(), mld.getBaseDictionary())
    path = str(tmp_path / 'dict.pkl')
    mld.save(path)
    back = MultilevelDictionary.restore(path)
    assert all(np.array_equal(x, y) for x, y in zip(back.dictionaries, mld.dictionaries))


@pytest.mark.gpu
def test_generated_signal_is_recovered_by_the_hierarchical_encoder():
    """End to end on synthetic data of the reference's kind: events -> signal -> 2-level encode on the GPU;
    the encoder explains the signal mostly with level-1 atoms."""
    from hsc_amd.dataset import SignalGenerator
    from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit, HierarchicalConvolutionalSparseCoder
    mld = _generate('two_level')
    rs = np.random.RandomState(5)
    gen = SignalGenerator(mld, [1e-9, 0.004], rng=rs)           # level-1 events only
    events = gen.generateEvents(4096)
    x = gen.generateSignalFromEvents(events, nbSamples=4096)
    hcsc = HierarchicalConvolutionalSparseCoder(mld, HierarchicalConvolutionalMatchingPursuit(method='cmp'))
    coefficients, residual = hcsc.encode(x, toleranceSnr=[20.0, 20.0], nbBlocks=1, singletonWeight=0.5)
    snr = 10 * np.log10(np.sum(np.square(x.astype(np.float64))) / np.sum(np.square(residual.astype(np.float64))))
    assert snr >= 12.0            # each level meets its own target in ITS input space (modeling.py:1439-1442)
    assert coefficients[1].nnz > 0 and coefficients[1].shape == (4096, mld.withSingletonBases().counts[1])
    rec = hcsc.reconstruct(coefficients)
    assert np.allclose(rec + residual, x, atol=1e-4)


# ---- dictionary files (hsc/dataset.py:378-396) ------------------------------------------------------------------------
def test_restore_dictionary_saved_by_the_reference(tmp_path):
    """tests/golden/mld_reference.pkl was written by the REAL reference's MultilevelDictionary (class path `hsc.dataset`,
    pickle protocol 2, tools/make_golden.py mldpkl): it restores into this package's container with the same content."""
    import os
    import golden_util as gu
    from hsc_amd.dataset import MultilevelDictionary
    z = gu.load('hsc_small.npz')
    m = MultilevelDictionary.restore(os.path.join(gu.GOLDEN, 'mld_reference.pkl'))
    assert isinstance(m, MultilevelDictionary) and m.getNbLevels() == 3 and m.hasSingletonBases
    assert np.array_equal(m.counts, z['counts']) and np.array_equal(m.countsNoSingletons, z['countsNoSingletons'])
    for l in range(3):
        assert np.array_equal(m.getRawDictionary(l), z['single_raw%d' % l])
        assert np.array_equal(m.getMultiscaleDictionaries()[l], z['single_rep%d' % l])
    # save / restore round trip, and the reference's extension rule
    path = os.path.join(str(tmp_path), 'again.p')
    m.save(path)
    m2 = MultilevelDictionary.restore(path)
    assert all(np.array_equal(a, b) for a, b in zip(m.dictionaries, m2.dictionaries))
    with pytest.raises(Exception, match='Unsupported format'):
        m.save(os.path.join(str(tmp_path), 'dict.npz'))
    with pytest.raises(Exception, match='Unsupported format'):
        MultilevelDictionary.restore(os.path.join(str(tmp_path), 'dict.json'))


def test_restore_python2_style_pickle(tmp_path):
    """What Python 2's cPickle actually writes: byte strings as (SHORT_)BINSTRING opcodes -- numpy array buffers among
    them -- and `copy_reg` / `__builtin__` module names.  Emulated by a pickler that emits those opcodes."""
    import os
    import pickle
    import struct
    import golden_util as gu
    from hsc_amd.dataset import MultilevelDictionary

    class Py2Pickler(pickle._Pickler):
        def save_bytes(self, obj):
            n = len(obj)
            self.write((b'U' + bytes([n]) if n < 256 else b'T' + struct.pack('<i', n)) + obj)
            self.memoize(obj)
        dispatch = dict(pickle._Pickler.dispatch)
        dispatch[bytes] = save_bytes

    src = MultilevelDictionary.restore(os.path.join(gu.GOLDEN, 'mld_reference.pkl'))
    path = os.path.join(str(tmp_path), 'py2.pkl')
    with open(path, 'wb') as f:
        Py2Pickler(f, protocol=2).dump(src)
    raw = open(path, 'rb').read()
    raw = raw.replace(b'chsc_amd.dataset\n', b'chsc.dataset\n').replace(b'ccopyreg\n', b'ccopy_reg\n')
    open(path, 'wb').write(raw)
    with pytest.raises(Exception):
        pickle.loads(raw)                                  # a plain load cannot: unknown module / undecodable byte strings
    m = MultilevelDictionary.restore(path)
    assert all(np.array_equal(a, b) for a, b in zip(src.dictionaries, m.dictionaries))
    assert all(np.array_equal(a, b) for a, b in zip(src.representations, m.representations))
    assert np.array_equal(src.counts, m.counts) and m.hasSingletonBases
