"""The round-parallel loop (csrc/hscmp_rp.h: a 1024-thread workgroup per signal applies the atoms of a blocked
selection round side by side) against the one-atom-at-a-time loop (iterate_kernel) and the CPU oracle, bit for bit.

HSCMP_RP=1 / 0 force either loop (read at every launch); by default the dispatcher picks the round-parallel form for
blocked selection when the batch leaves CUs idle, so the rest of the -m gpu suite (fuzz sweeps, edges, goldens) runs it
wherever it applies as well."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _encode(xs, D, forced, monkeypatch, **kw):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    monkeypatch.setenv('HSCMP_RP', forced)
    cmp = ConvolutionalMatchingPursuit()
    res = cmp.computeCoefficientsBatch(xs, D, **kw)
    return res


def _same(a, b, B):
    for i in range(B):
        for u, v in zip(a.events[i], b.events[i]):
            assert np.array_equal(u, v), i
        assert np.array_equal(a.residuals[i], b.residuals[i]), i
    assert np.array_equal(a.stats, b.stats)


CASES = [
    # (K, W, T, nbBlocks, kwargs)
    (32, 32, 4096, 10, dict(nbNonzeroCoefs=64)),
    (32, 32, 4096, 'auto', dict(nbNonzeroCoefs=100)),
    (128, 32, 8192, 10, dict(toleranceSnr=25.0)),
    (70, 16, 3000, 7, dict(toleranceSnr=15.0, nbNonzeroCoefs=200)),
    (256, 64, 8192, 10, dict(toleranceSnr=20.0)),
    (96, 64, 4100, 'auto', dict(nbNonzeroCoefs=120)),
    (33, 8, 1000, 3, dict(toleranceSnr=30.0)),
    (48, 60, 5000, 5, dict(nbNonzeroCoefs=80)),
]


@pytest.mark.parametrize('case', range(len(CASES)))
@pytest.mark.parametrize('weighted', [False, True])
def test_round_parallel_vs_sequential_vs_oracle(case, weighted, monkeypatch):
    import hsc_amd.synth as synth
    from oracle import hsc_oracle as orc
    K, W, T, nb, kw = CASES[case]
    kw = dict(kw, nbBlocks=nb)
    D = synth.make_dictionary(K, W, seed=30 + case)
    xs = np.stack([synth.make_signal(D, T, i, kind='planted' if i % 3 else 'noise', nb_atoms=max(8, T // (3 * W)), noise=0.05, seed=31 + case)
                   for i in range(5)])
    xs[4, : 2 * W] *= 6.0                                       # energy piled at the left end: atoms whose windows cross it
    xs[3, -2 * W:] *= 6.0
    if weighted:
        kw['weights'] = np.random.RandomState(case).uniform(0.6, 1.0, size=K).astype(np.float32)
    rp = _encode(xs, D, '1', monkeypatch, **kw)
    # (built for dictionaries of 2, 4 or 8 chunks of 8 taps; other widths keep the sequential loop)
    assert rp.variant.endswith('_rp') == ((W + 7) // 8 in (2, 4, 8)), rp.variant
    seq = _encode(xs, D, '0', monkeypatch, **kw)
    assert not seq.variant.endswith('_rp'), seq.variant
    _same(rp, seq, xs.shape[0])
    for i in (0, 3, 4):
        coef, r, info = orc.cmp_encode(xs[i], D, **kw)
        t, k, c = rp.events[i]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), (case, i)
        assert np.array_equal(np.squeeze(rp.residuals[i]), r), (case, i)
        assert rp.stop_reasons()[i] == info['stop']


@pytest.mark.parametrize('W', [12, 16, 32])
def test_round_whose_interference_filter_is_skipped(W, monkeypatch):
    """Two strong atoms closer than W on either side of a block boundary, nothing else: the candidates of the round have no
    qualifying gap, modeling.py:951-957 skips the filter, and the overlapping atoms must be applied one after the other
    with the energies of their turn."""
    import hsc_amd.synth as synth
    from oracle import hsc_oracle as orc
    K, T = 24, 64 * W
    D = synth.make_dictionary(K, W, seed=50 + W)
    xs = np.zeros((3, T), dtype=np.float32)
    for i, (gap, nbk) in enumerate([(3, 2), (W - 1, 4), (W // 2, 8)]):
        edge = (T // nbk) * (nbk // 2)                         # a block boundary of the un-shifted rounds
        for p, k, c in ((edge - gap // 2 - 1, 3, 2.5), (edge + gap - gap // 2 - 1, 7, -1.75)):
            lo = p - (W - 1) // 2
            xs[i, lo:lo + W] += c * D[k]
    for i, nbk in enumerate([2, 4, 8]):
        kw = dict(nbBlocks=nbk, nbNonzeroCoefs=12)
        rp = _encode(xs[i:i + 1], D, '1', monkeypatch, **kw)
        assert rp.variant.endswith('_rp')
        seq = _encode(xs[i:i + 1], D, '0', monkeypatch, **kw)
        _same(rp, seq, 1)
        coef, r, info = orc.cmp_encode(xs[i], D, **kw)
        t, k, c = rp.events[0]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), (W, i)
        assert np.array_equal(np.squeeze(rp.residuals[0]), r)


def test_round_parallel_resumes_after_event_list_growth(monkeypatch):
    import hsc_amd.synth as synth
    D = synth.make_dictionary(64, 32, seed=60)
    xs = synth.make_batch(D, 8192, 0, 3, kind='planted', nb_atoms=120, noise=0.02, seed=60)
    kw = dict(nbBlocks=10, toleranceSnr=30.0)
    full = _encode(xs, D, '1', monkeypatch, **kw)
    short = _encode(xs, D, '1', monkeypatch, maxEvents=4, **kw)   # grown in place, the loop resumed several times
    _same(full, short, 3)


def test_config5_level0_dims_three_runs(monkeypatch):
    """BASELINE config 5, level 0 (128 atoms x 32 taps, 65536 samples, toleranceSnr 30, nbBlocks=10; ~17 k atoms per
    signal): both loops on every signal, the round-parallel one three times, bit for bit."""
    import hsc_amd.synth as synth
    D = synth.make_dictionary(128, 32, seed=5)
    B, T = 24, 65536
    xs = np.stack([synth.make_signal(D, T, i, kind='planted', nb_atoms=T // 6, noise=0.02, seed=5) for i in range(B)])
    kw = dict(nbBlocks=10, toleranceSnr=30.0)
    seq = _encode(xs, D, '0', monkeypatch, **kw)
    assert int(seq.stats[:, 4].min()) > 3000                    # thousands of atoms per signal
    for _ in range(3):
        rp = _encode(xs, D, '1', monkeypatch, **kw)
        assert rp.variant.endswith('_rp')
        _same(rp, seq, B)


def _level_problem(seed, W, F, K, T, dense_rows=0):
    """A level >= 1 shaped problem (sparse input rows, sparse dictionary with singleton atoms in front), float64."""
    rs = np.random.RandomState(seed)
    D = np.zeros((K, W, F))
    for k in range(K):
        for _ in range(int(rs.randint(1, 6))):
            D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
        D[k] /= np.sqrt(np.sum(np.square(D[k])))
    S = np.zeros((F, W, F)); S[np.arange(F), (W - 1) // 2, np.arange(F)] = 1.0
    D = np.concatenate((S, D), axis=0)
    x = np.zeros((T, F))
    for _ in range(int(rs.randint(T // 12, T // 4))):
        x[rs.randint(0, T), rs.randint(0, F)] = rs.uniform(0.3, 2.0) * rs.choice([-1.0, 1.0])
    for t in rs.randint(0, T, size=dense_rows):
        n = min(F, int(rs.randint(9, 30)))
        x[t, rs.permutation(F)[:n]] = rs.standard_normal(n)            # a dense-ish row: its feature list overflows
    w = np.ones(D.shape[0]); w[:F] = rs.uniform(0.7, 0.95)
    return x, D, w


LEVEL_CASES = [
    # (W, F, K, T, nbBlocks, dense rows)
    (16, 48, 20, 1400, 10, 0),
    (17, 130, 12, 1100, 10, 0),
    (33, 48, 16, 2000, 10, 0),
    (65, 130, 10, 1800, 7, 0),
    (9, 12, 6, 700, 'auto', 0),
    (16, 48, 20, 1500, 10, 3),
    (33, 130, 8, 1600, 4, 2),
    (5, 30, 8, 300, 3, 0),
]


@pytest.mark.parametrize('case', range(len(LEVEL_CASES)))
def test_level_shaped_round_parallel_vs_sequential_vs_oracle(case, monkeypatch):
    from oracle import hsc_oracle as orc
    W, F, K, T, nb, dense = LEVEL_CASES[case]
    probs = [_level_problem(7000 + 10 * case + i, W, F, K, T, dense) for i in range(3)]
    D, w = probs[0][1], probs[0][2]
    xs = np.stack([p[0] for p in probs])
    xs[1, :W] *= 3.0; xs[2, -W:] *= 3.0                          # atoms at the signal ends
    kw = dict(toleranceSnr=22.0, nbNonzeroCoefs=400, nbBlocks=nb, weights=w, minCoefficients=None)
    rp = _encode(xs, D, '1', monkeypatch, **kw)
    assert rp.variant.endswith('_rp'), rp.variant
    seq = _encode(xs, D, '0', monkeypatch, **kw)
    assert not seq.variant.endswith('_rp'), seq.variant
    _same(rp, seq, xs.shape[0])
    coef, r, info = orc.cmp_encode(xs[1], D, maxEvents=2000, **kw)
    if info['stop'] != 'capacity':
        t, k, c = rp.events[1]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), case
        assert np.array_equal(rp.residuals[1], r), case
        assert rp.stop_reasons()[1] == info['stop']


def test_rounds_of_more_than_64_blocks(monkeypatch):
    """nbBlocks='auto' on a long signal: 256 blocks per round -- the filters run over index lists by the whole workgroup
    (rp_block_select), the bookkeeping prefix in chunks of 64 atoms."""
    import hsc_amd.synth as synth
    from oracle import hsc_oracle as orc
    D = synth.make_dictionary(32, 16, seed=70)
    T = 16384
    xs = np.stack([synth.make_signal(D, T, i, kind='planted', nb_atoms=T // 40, noise=0.03, seed=70) for i in range(3)])
    for kw in (dict(nbBlocks='auto', nbNonzeroCoefs=700), dict(nbBlocks='auto', toleranceSnr=18.0), dict(nbBlocks=200, nbNonzeroCoefs=150)):
        rp = _encode(xs, D, '1', monkeypatch, **kw)
        assert rp.variant.endswith('_rp'), rp.variant
        seq = _encode(xs, D, '0', monkeypatch, **kw)
        _same(rp, seq, 3)
        coef, r, info = orc.cmp_encode(xs[2], D, **kw)
        t, k, c = rp.events[2]
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c'])
        assert rp.stop_reasons()[2] == info['stop']


def test_more_signals_than_compute_units(monkeypatch):
    """A batch larger than the chip: the workgroups of the round-parallel loop run in several waves of residency."""
    import hsc_amd.synth as synth
    D = synth.make_dictionary(48, 16, seed=71)
    B, T = 600, 2048
    xs = synth.make_batch(D, T, 0, B, kind='planted', nb_atoms=40, noise=0.02, seed=71)
    kw = dict(nbBlocks=6, toleranceSnr=25.0)
    rp = _encode(xs, D, '1', monkeypatch, **kw)
    assert rp.variant.endswith('_rp')
    seq = _encode(xs, D, '0', monkeypatch, **kw)
    _same(rp, seq, B)
