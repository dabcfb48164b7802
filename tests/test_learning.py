"""Dictionary learners around the GPU engine (reference hsc/modeling.py:82-147,
265-655).  The k-means learner's correlate + arg-max step runs on the GPU (hscmp_assign_windows); the
rest is host logic that draws random numbers in the reference's order, so under the seeds of
tests/golden/learn_small.npz (written by tools/make_golden.py from the real reference) the learned
dictionaries must be reproduced."""
import numpy as np
import pytest

import golden_util as gu


def _oracle_assign(self, windows, D):
    """CPU stand-in of the GPU step for the host-logic tests: oracle 'valid' correlation + flat arg-max."""
    from oracle import hsc_oracle as orc
    w3 = windows.reshape((windows.shape[0], windows.shape[1], -1))
    D3 = D.reshape((D.shape[0], D.shape[1], -1))
    dt = np.result_type(w3.dtype, D3.dtype)
    t = np.zeros(len(w3), dtype=np.int64); k = np.zeros(len(w3), dtype=np.int64)
    for n in range(len(w3)):
        ip = orc.convolve1d(np.ascontiguousarray(w3[n], dtype=dt), np.ascontiguousarray(D3, dtype=dt), padding='valid')
        o = int(np.argmax(np.abs(ip).reshape(-1)))
        t[n], k[n] = o // ip.shape[1], o % ip.shape[1]
    return t, k


def _train(name, monkeypatch=None):
    from hsc_amd.learning import ConvolutionalDictionaryLearner
    sig, k, w, seed, kw = gu.LEARN_CASES[name]
    if monkeypatch is not None:
        monkeypatch.setattr(ConvolutionalDictionaryLearner, '_assign', _oracle_assign)
    np.random.seed(seed)
    return ConvolutionalDictionaryLearner(k, w, algorithm='kmean').train(gu.learn_signal(sig), **kw)


@pytest.mark.parametrize('name', sorted(gu.LEARN_CASES))
def test_kmean_host_logic_matches_reference(name, monkeypatch):
    exp = gu.load('learn_small.npz')[name + '__D']
    D = _train(name, monkeypatch)
    assert D.shape == exp.shape and D.dtype == exp.dtype
    assert np.array_equal(D, exp)


def test_samples_learner_and_window_helpers():
    from hsc_amd.learning import ConvolutionalDictionaryLearner, extractWindows, extractWindowsBatch, extractRandomWindows
    z = gu.load('learn_small.npz')
    data = gu.learn_signal('sparse_2d')
    np.random.seed(7)
    D = ConvolutionalDictionaryLearner(5, 9, algorithm='samples').train(data, avoidSingletons=True)
    assert np.array_equal(D, z['samples_2d__D'])
    assert np.all(np.sum(D != 0.0, axis=(1, 2)) > 1) and np.allclose(np.sum(np.square(D), axis=(1, 2)), 1.0)
    x = np.arange(40.0)
    assert np.array_equal(extractWindows(x, np.array([3, 10]), 4), [[3, 4, 5, 6], [10, 11, 12, 13]])
    assert np.array_equal(extractWindows(x, np.array([3, 10]), 4, centered=True), [[2, 3, 4, 5], [9, 10, 11, 12]])
    assert np.array_equal(extractWindows(x, np.array([3, 10]), 5, centered=True), [[1, 2, 3, 4, 5], [8, 9, 10, 11, 12]])
    b = np.stack([x, x + 100])
    assert np.array_equal(extractWindowsBatch(b, np.array([1, 2]), 3), [[1, 2, 3], [102, 103, 104]])
    assert extractWindowsBatch(b[:, :, np.newaxis], np.array([1, 2]), 3).shape == (2, 3, 1)
    w = extractRandomWindows(x, 7, 6, rng=np.random.RandomState(0))
    assert w.shape == (7, 6) and np.all(np.diff(w, axis=1) == 1)
    with pytest.raises(NotImplementedError):
        ConvolutionalDictionaryLearner(3, 4, algorithm='nmf').train(x)
    with pytest.raises(Exception):
        ConvolutionalDictionaryLearner(3, 4, algorithm='bogus').train(x)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(gu.LEARN_CASES))
def test_assign_windows_vs_oracle_and_reference(name):
    """hscmp_assign_windows against the oracle (bit-exact coefficient, exact indices) and the reference's own
    convolve1d_batch + arg-max on the same windows."""
    from hsc_amd import _native
    from hsc_amd.learning import extractWindows
    from oracle import hsc_oracle as orc
    z = gu.load('learn_small.npz')
    sig, k, w, seed, kw = gu.LEARN_CASES[name]
    data = gu.learn_signal(sig)
    D = z[name + '__D']
    windows = extractWindows(data, z[name + '__win_idx'], 2 * w)
    dt = np.result_type(windows.dtype, D.dtype)
    eng = _native.Engine(0)
    eng.set_dictionary(np.ascontiguousarray(D.reshape((D.shape[0], D.shape[1], -1)), dtype=dt))
    t, kk, c = eng.assign_windows(windows.astype(dt))
    w3 = windows.reshape((windows.shape[0], windows.shape[1], -1)).astype(dt)
    for n in range(len(w3)):
        ip = orc.convolve1d(np.ascontiguousarray(w3[n]), np.ascontiguousarray(D.reshape((D.shape[0], D.shape[1], -1)), dtype=dt), padding='valid')
        o = int(np.argmax(np.abs(ip).reshape(-1)))
        assert (t[n], kk[n]) == (o // ip.shape[1], o % ip.shape[1])
        assert c[n] == ip[t[n], kk[n]]
    assert np.array_equal(t, z[name + '__assign_t']) and np.array_equal(kk, z[name + '__assign_k'])
    assert np.allclose(c, z[name + '__assign_c'], rtol=1e-5, atol=1e-6)
    eng.close()


@pytest.mark.gpu
def test_assign_windows_long_windows_and_ties():
    """Windows too long for LDS take the global-memory path; equal maxima resolve in C order (position, then atom)."""
    from hsc_amd import _native
    from oracle import hsc_oracle as orc
    rs = np.random.RandomState(4)
    D = rs.standard_normal((5, 7, 3))
    windows = rs.standard_normal((3, 1500, 3))                     # 36 KB per window > 32 KB
    eng = _native.Engine(0)
    eng.set_dictionary(D)
    t, k, c = eng.assign_windows(windows)
    for n in range(3):
        ip = orc.convolve1d(windows[n], D, padding='valid')
        o = int(np.argmax(np.abs(ip).reshape(-1)))
        assert (t[n], k[n], c[n]) == (o // 5, o % 5, ip[o // 5, o % 5])
    # ties: two identical atoms, a periodic window
    D2 = np.stack([np.ones(4), np.ones(4), -np.ones(4)])[:, :, np.newaxis]
    eng.set_dictionary(D2)
    t, k, c = eng.assign_windows(np.ones((2, 9, 1)))
    assert t.tolist() == [0, 0] and k.tolist() == [0, 0] and c.tolist() == [4.0, 4.0]
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(gu.LEARN_CASES))
def test_kmean_on_gpu_matches_reference(name):
    exp = gu.load('learn_small.npz')[name + '__D']
    D = _train(name)
    assert D.shape == exp.shape and D.dtype == exp.dtype
    assert np.allclose(D, exp, rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_ksvd_on_gpu_matches_reference():
    """K-SVD (modeling.py:526-641) with the GPU matching pursuit as its coder: same dictionary as the reference
    under the same seed, atom by atom up to the sign the SVD leaves open."""
    from hsc_amd.learning import ConvolutionalDictionaryLearner
    exp = gu.load('learn_small.npz')['ksvd_1d__D']
    np.random.seed(3)
    D = ConvolutionalDictionaryLearner(6, 16, algorithm='ksvd').train(gu.learn_signal('planted_1d'), method='cmp', maxIterations=2,
                                                                     nbNonzeroCoefs=100, toleranceSnr=None)
    assert D.shape == exp.shape and D.dtype == exp.dtype
    assert np.allclose(np.sum(np.square(D), axis=1), 1.0, atol=1e-5)
    for a, b in zip(D, exp):
        assert min(np.max(np.abs(a - b)), np.max(np.abs(a + b))) <= 1e-4
