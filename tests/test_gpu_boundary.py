"""GPU tests of the boundary behaviours around the greedy loop: the stopCondition callback
(hsc/modeling.py:1155-1158) served by one launch per round, event-list capacity regrowth, resumable
launches (hscmp_continue), generic-vs-MFMA variant agreement, input immutability."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(T=2048, K=32, W=32, seed=3, n=24):
    import hsc_amd.synth as synth
    D = synth.make_dictionary(K, W, seed=seed)
    x = synth.make_signal(D, T, 0, kind='planted', nb_atoms=n, seed=seed)
    return x, D


def test_stop_condition_callback_equals_l0_rule():
    """With nbBlocks=1 a round is one atom, so `coefficients.nnz >= 7` after each round must stop
    exactly where nbNonzeroCoefs=7 does; the callback sees the reference's argument types."""
    import scipy.sparse
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    x, D = _setup()
    seen = []

    def stop(sequence, residual, coefficients):
        assert sequence.shape == (x.shape[0], 1) and residual.shape == (x.shape[0], 1)
        assert scipy.sparse.issparse(coefficients) and coefficients.shape == (x.shape[0], D.shape[0])
        seen.append(coefficients.nnz)
        return coefficients.nnz >= 7

    cmp1 = ConvolutionalMatchingPursuit()
    c1, r1 = cmp1.computeCoefficients(x, D, stopCondition=stop)
    cmp2 = ConvolutionalMatchingPursuit()
    c2, r2 = cmp2.computeCoefficients(x, D, nbNonzeroCoefs=7)
    assert seen == list(range(1, 8))
    assert all(np.array_equal(a, b) for a, b in zip(cmp1.lastResult.events[0], cmp2.lastResult.events[0]))
    assert np.array_equal(r1, r2) and (c1 != c2).nnz == 0
    assert cmp1.lastResult.stop_reasons() == ['callback'] and cmp2.lastResult.stop_reasons() == ['nnz']


@pytest.mark.parametrize('force_generic', [False, True])
@pytest.mark.parametrize('kw', [dict(nbNonzeroCoefs=20), dict(toleranceSnr=12.0, nbBlocks=4), dict(toleranceSnr=10.0, nbBlocks='auto')])
def test_event_capacity_regrowth(kw, force_generic, monkeypatch):
    """Event lists that are far too short are enlarged in place (hscmp_grow_events) and the loop resumed:
    same trace, residual and statistics as a run that never hit the limit -- rounds of several atoms
    (blocked selection) are never split."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    if force_generic:
        monkeypatch.setenv('HSCMP_FORCE_GENERIC', '1')
    x, D = _setup()
    a = ConvolutionalMatchingPursuit()
    a.computeCoefficientsBatch(x[np.newaxis], D, maxEvents=3, **kw)     # far too small: must grow (3 -> 12 -> 48 ...)
    b = ConvolutionalMatchingPursuit()
    b.computeCoefficientsBatch(x[np.newaxis], D, maxEvents=8192, **kw)
    assert len(b.lastResult.events[0][0]) > 12
    assert all(np.array_equal(u, v) for u, v in zip(a.lastResult.events[0], b.lastResult.events[0]))
    assert np.array_equal(a.lastResult.residuals, b.lastResult.residuals)
    assert np.array_equal(a.lastResult.stats[:, :5], b.lastResult.stats[:, :5])
    assert (a.lastResult.coefficients[0] != b.lastResult.coefficients[0]).nnz == 0


def test_event_capacity_regrowth_with_callback():
    """Growth in stopCondition mode (one launch per round): the callback is not evaluated twice on the same state."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    x, D = _setup()
    seen = []

    def stop(sequence, residual, coefficients):
        seen.append(coefficients.nnz)
        return coefficients.nnz >= 9

    a = ConvolutionalMatchingPursuit()
    a.computeCoefficientsBatch(x[np.newaxis], D, stopCondition=stop, maxEvents=2)
    assert seen == list(range(1, 10))


def test_resumed_launches_equal_one_launch():
    """max_rounds-limited launches + hscmp_continue reproduce the single-launch trace bit for bit
    (the kernel state -- segment maxima, Bloom filter, edge flags -- is rebuilt on every launch)."""
    from hsc_amd import _native
    x, D = _setup(T=1024, K=16, W=16, n=10)
    # atoms near both edges exercise the edge-row flags across launches
    x = x.copy(); x[:10] += 1.5 * D[3][-10:]; x[-12:] -= 2.0 * D[5][:12]
    eng = _native.Engine(0)
    eng.set_dictionary(D)
    eps = float(np.finfo(np.float32).eps)
    full = _native.make_params(nbNonzeroCoefs=30, eps=eps, maxEvents=128)
    eng.encode_batch(x[np.newaxis, :, np.newaxis], full)
    t0, k0, c0 = [a[0].copy() for a in eng.fetch_events()]
    st0 = eng.fetch_stats()[0].copy()
    r0 = eng.fetch_residual().copy()
    step = _native.make_params(nbNonzeroCoefs=30, eps=eps, maxEvents=128, maxRounds=3)
    eng.encode_batch(x[np.newaxis, :, np.newaxis], step)
    for _ in range(40):
        if eng.fetch_stats()[0, _native.STAT_STOP] != _native.STOP_RUNNING:
            break
        eng.continue_rounds(3)
    t1, k1, c1 = [a[0] for a in eng.fetch_events()]
    st1 = eng.fetch_stats()[0]
    n = st0[_native.STAT_EVENTS]
    assert st1[_native.STAT_EVENTS] == n and st1[_native.STAT_STOP] == st0[_native.STAT_STOP]
    assert np.array_equal(t0[:n], t1[:n]) and np.array_equal(k0[:n], k1[:n]) and np.array_equal(c0[:n], c1[:n])
    assert np.array_equal(r0, eng.fetch_residual())


def test_generic_and_mfma_variants_agree():
    """Same inputs through the VALU kernels (HSCMP_FORCE_GENERIC) and the MFMA kernels: identical."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    x, D = _setup(T=4096, K=64, W=32, n=40)
    w = np.ones(64, dtype=np.float32); w[:10] = 0.5
    outs = []
    for force in (False, True):
        if force:
            os.environ['HSCMP_FORCE_GENERIC'] = '1'
        try:
            cmp = ConvolutionalMatchingPursuit()
            cmp.computeCoefficients(x, D, nbNonzeroCoefs=40, weights=w, nbBlocks=4)
            outs.append((cmp.lastResult.events[0], cmp.lastResult.residuals[0].copy(), cmp.lastResult.variant))
        finally:
            os.environ.pop('HSCMP_FORCE_GENERIC', None)
    assert outs[0][2].startswith('mfma') and outs[1][2].startswith('generic')
    assert all(np.array_equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    assert np.array_equal(outs[0][1], outs[1][1])


def test_inputs_are_not_mutated():
    """modeling.py:1071: the residual is a copy; sequence and D stay untouched."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    x, D = _setup()
    x0, D0 = x.copy(), D.copy()
    ConvolutionalMatchingPursuit().computeCoefficients(x, D, nbNonzeroCoefs=10)
    assert np.array_equal(x, x0) and np.array_equal(D, D0)


def test_chained_level_energy_with_crowded_partial_sums():
    """Level chaining takes the signal energy from the previous level's slots.  With more cells per pinned partial
    sum than its private lists hold (here 30000 cells on 128 of the 256 partial sums) it walks the per-row feature
    lists instead: the energy must still equal the oracle's pinned sum over the dense input, bit for bit."""
    from hsc_amd import _native
    from oracle import hsc_oracle as orc
    rs = np.random.RandomState(8)
    T = 30000
    x = (rs.uniform(0.5, 2.0, size=T) * rs.choice([-1.0, 1.0], size=T)).astype(np.float32)
    D0 = np.array([[1.0], [0.5]], dtype=np.float32)                 # one tap: every sample becomes one event of atom 0
    e0 = _native.Engine(0); e0.set_dictionary(D0.reshape(2, 1, 1))
    p0 = _native.make_params(nbNonzeroCoefs=T, eps=float(np.finfo(np.float32).eps), maxEvents=T + 8)
    e0.encode_batch(x.reshape(1, T, 1), p0)
    st = e0.fetch_stats()
    assert st[0][_native.STAT_SLOTS] == T
    D1 = np.zeros((3, 4, 2)); D1[0, 1, 0] = 1.0; D1[1, 1, 1] = 1.0; D1[2, 0, 0] = D1[2, 3, 0] = np.sqrt(0.5)
    e1 = _native.Engine(0); e1.set_dictionary(D1, np.ones(3))
    p1 = _native.make_params(nbNonzeroCoefs=5, eps=float(np.finfo(np.float32).eps), maxEvents=64)
    e1.encode_batch_from_level(e0, 0, 1, 1e-16, p1)
    assert e1.last_variant().startswith('dictlist_init+dictlist_loop')
    dense = np.zeros((T, 2)); dense[:, 0] = x.astype(np.float64)     # what the scatter produced
    energies = e1.fetch_energies()
    assert energies[0][0] == orc.energy(dense.reshape(-1))
    # and the pursuit that follows is the oracle's
    coef, res, info = orc.cmp_encode(dense, D1, nbNonzeroCoefs=5, weights=np.ones(3))
    ev = e1.fetch_events(); n = e1.fetch_stats()[0][_native.STAT_EVENTS]
    assert np.array_equal(ev[0][0][:n], info['t']) and np.array_equal(ev[2][0][:n], info['c'])
    e0.close(); e1.close()
