"""Seeded random sweep (-m gpu) of shapes, dtypes, stop rules and selection modes through the single-signal
entry point, bit for bit against the CPU oracle.  The shapes are drawn so that every dispatch path is hit:
MFMA (F = 1, T >= 3W-2), generic (short signals, F > 1 with dense dictionaries), sparse multi-feature paths
(sparse dictionaries with singletons)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import os
N_CASES = int(os.environ.get("HSCMP_FUZZ_CASES", "1600"))
OFFSET = int(os.environ.get("HSCMP_FUZZ_OFFSET", "0"))           # soak runs explore other seeds


def _draw(i):
    rs = np.random.RandomState(9000 + i)
    dtype = np.float32 if rs.rand() < 0.5 else np.float64
    family = i % 4
    if family == 0:                                           # single feature, MFMA-sized
        W = int(rs.choice([3, 4, 7, 8, 16, 17, 31, 32, 40]))
        K = int(rs.choice([1, 2, 5, 16, 31, 32, 33, 48, 70]))
        T = int(rs.randint(3 * W, 40 * W + 50))
        F = 1
    elif family == 1:                                         # single feature, short / odd
        W = int(rs.randint(1, 24)); K = int(rs.randint(1, 12)); T = int(rs.randint(max(2, W // 2), 3 * W + 4)); F = 1
    elif family == 2:                                         # multi-feature, dense dictionary
        W = int(rs.randint(2, 12)); K = int(rs.randint(1, 10)); T = int(rs.randint(2 * W, 30 * W)); F = int(rs.randint(2, 7))
    else:                                                     # multi-feature, sparse dictionary + singletons
        W = int(rs.randint(2, 20)); F = int(rs.randint(2, 40)); K = int(rs.randint(1, 16)); T = int(rs.randint(3 * W, 60 * W))
    if family == 3:
        D = np.zeros((K, W, F), dtype=dtype)
        for k in range(K):
            for _ in range(int(rs.randint(1, 5))):
                D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
            D[k] /= np.sqrt(np.sum(np.square(D[k])))
        S = np.zeros((F, W, F), dtype=dtype)
        S[np.arange(F), (W - 1) // 2, np.arange(F)] = 1.0
        D = np.concatenate((S, D), axis=0)
        x = np.zeros((T, F), dtype=dtype)
        for _ in range(max(1, T // 12)):
            x[rs.randint(0, T), rs.randint(0, F)] = rs.uniform(0.3, 2.0) * rs.choice([-1.0, 1.0])
    else:
        shape = (K, W) if F == 1 else (K, W, F)
        D = rs.standard_normal(shape).astype(dtype)
        D /= np.sqrt(np.sum(np.square(D), axis=tuple(range(1, D.ndim)), keepdims=True))
        x = rs.standard_normal((T,) if F == 1 else (T, F)).astype(dtype)
        if rs.rand() < 0.5:                                   # planted structure on top of weak noise
            x *= dtype(0.05)
            D3 = D.reshape((K, W, -1)); x2 = x.reshape((T, -1))
            for _ in range(max(1, T // (2 * W))):
                k = rs.randint(0, K); t = rs.randint(0, max(1, T - W)); n = min(W, T - t)
                x2[t:t + n] += (rs.uniform(0.5, 2.0) * D3[k][:n]).astype(dtype)
    kw = {}
    rule = rs.randint(0, 4)
    if rule == 0:
        kw['nbNonzeroCoefs'] = int(rs.randint(1, 40))
    elif rule == 1:
        kw['toleranceSnr'] = float(rs.uniform(3.0, 25.0)); kw['nbNonzeroCoefs'] = 60
    elif rule == 2:
        kw['toleranceResidualScale'] = float(rs.uniform(0.2, 1.0)) * float(np.max(np.abs(x))); kw['nbNonzeroCoefs'] = 60
    else:
        kw['nbNonzeroCoefs'] = int(rs.randint(1, 25)); kw['toleranceSnr'] = 30.0
    mode = rs.randint(0, 3)
    if mode == 1 and T >= 8:
        kw['nbBlocks'] = int(rs.randint(2, min(9, T // 2)))
    elif mode == 2:
        kw['nbBlocks'] = 'auto'
    if rs.rand() < 0.3:
        w = rs.uniform(0.5, 1.0, size=D.shape[0]).astype(dtype)
        kw['weights'] = w
    if rs.rand() < 0.2:
        kw['minCoefficients'] = None
    return x, D, kw


# seeds beyond the default range that exercise the stale reflected sample of row T-1 (even W: an atom at T-1-W
# changes r[T-1-W/2] without re-correlating row T-1, see edge_window_value in csrc/hscmp_mfma.h)
STALE_EDGE_SEEDS = [401, 2376, 2460, 2768, 3004, 3136, 4072, 4128, 4188, 4272, 4896, 5504]


# a pursuit that diverges (filter longer than a third of the signal) until the float32 residual overflows: every score of a row
# is NaN then, and the per-row arg-max over atoms must still name an atom (a soak run found the wild index it used to leave)
DIVERGING_SEEDS = [301678]


@pytest.mark.parametrize('i', list(range(N_CASES)) + [q for q in STALE_EDGE_SEEDS + DIVERGING_SEEDS if q >= N_CASES])
def test_random_configuration_vs_oracle(i):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from hsc_amd._native import HscmpError
    from oracle import hsc_oracle as orc
    x, D, kw = _draw(i + OFFSET if i < N_CASES else i)
    okw = dict(kw)
    coef, res, info = orc.cmp_encode(x, D, maxEvents=1 << 17, **okw)
    if info['stop'] == 'capacity':
        # more than 131072 selections: either a pursuit that cannot terminate (the engine then says so once its own,
        # larger bound is reached -- pinned by test_gpu_edges.py) or one that converges later than the oracle was
        # allowed to follow; nothing to compare, it must just end
        try:
            ConvolutionalMatchingPursuit().computeCoefficients(x, D, **kw)
        except HscmpError as ex:
            assert 'does not converge' in str(ex)
        return
    cmp = ConvolutionalMatchingPursuit()
    diverged = not np.all(np.isfinite(res))
    try:
        coefficients, residual = cmp.computeCoefficients(x, D, **kw)
    except HscmpError as ex:
        # only a diverging pursuit may end like this (after the overflow the stop tests see inf / NaN, whose
        # comparisons are not pinned: the oracle happened to stop, the engine ran into its event bound)
        assert diverged and 'does not converge' in str(ex)
        return
    t, k, c = cmp.lastResult.events[0]
    if diverged:
        # a diverging pursuit (filters longer than the signal: the reflect-padded re-correlation feeds on itself until
        # the residual overflows): identical up to the overflow, the order of inf / NaN comparisons after it is not pinned
        huge = np.where(~(np.abs(info['c'].astype(np.float64)) < 1e300 if info['c'].dtype == np.float64 else np.abs(info['c']) < 1e30))[0]
        n = min(len(t), len(info['t']), int(huge[0]) if len(huge) else 1 << 30) - 2
        assert n > 0 and np.array_equal(t[:n], info['t'][:n]) and np.array_equal(k[:n], info['k'][:n]) and np.array_equal(c[:n], info['c'][:n])
        return
    tag = (i, cmp.lastResult.variant, x.shape, D.shape, {a: b for a, b in kw.items() if a != 'weights'})
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']), tag
    assert np.array_equal(c, info['c']), tag
    assert np.array_equal(residual, res), tag
    assert (coefficients != coef).nnz == 0, tag
    assert cmp.lastResult.stop_reasons()[0] == info['stop'], tag
    if i % 3 == 0:
        # the same run squeezed through event lists that start far too short (grown in place, loop resumed)
        small = ConvolutionalMatchingPursuit()
        r2 = small.computeCoefficientsBatch(x[np.newaxis], D, maxEvents=2, **kw)
        t2, k2, c2 = r2.events[0]
        assert np.array_equal(t2, t) and np.array_equal(k2, k) and np.array_equal(c2, c), tag
        assert np.array_equal(np.squeeze(r2.residuals[0]), np.squeeze(residual)) and r2.stop_reasons()[0] == info['stop'], tag
    if i % 5 == 0:
        # stopCondition served by one launch per round (hscmp_continue): stop after a few rounds, as the oracle's maxRounds
        nrounds = 1 + i % 4
        seen = []

        def stop(sequence, residual_, coefficients_):
            seen.append(1)
            return len(seen) >= nrounds

        cb = ConvolutionalMatchingPursuit()
        cb.computeCoefficients(x, D, stopCondition=stop, **kw)
        t3, k3, c3 = cb.lastResult.events[0]
        coef3, res3, info3 = orc.cmp_encode(x, D, maxEvents=1 << 17, maxRounds=nrounds, **kw)
        assert np.array_equal(t3, info3['t']) and np.array_equal(k3, info3['k']) and np.array_equal(c3, info3['c']), (tag, 'callback', nrounds)


N_BATCHES = int(os.environ.get("HSCMP_FUZZ_BATCHES", "160"))


@pytest.mark.parametrize('i', range(N_BATCHES))
def test_random_batch_vs_oracle(i):
    """Batches of related signals (scaled, reversed, noisier, all-zero, truncated energy) through the batch entry
    point: every signal must equal its own single-signal oracle run -- signals converge at different times."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from hsc_amd._native import HscmpError
    from oracle import hsc_oracle as orc
    x, D, kw = _draw(20000 + i)
    rs = np.random.RandomState(777 + i)
    xs = np.stack([x, 0.5 * x, x[::-1].copy(), np.zeros_like(x), x + (0.1 * rs.standard_normal(x.shape)).astype(x.dtype),
                   (x * (np.arange(x.shape[0]) < x.shape[0] // 2).reshape((-1,) + (1,) * (x.ndim - 1))).astype(x.dtype)])
    refs = [orc.cmp_encode(xs[b], D, maxEvents=1 << 17, **kw) for b in range(xs.shape[0])]
    if any(r[2]['stop'] == 'capacity' or not np.all(np.isfinite(r[1])) for r in refs):
        return                                              # non-terminating / diverging member: covered by the single-signal sweep
    cmp = ConvolutionalMatchingPursuit()
    res = cmp.computeCoefficientsBatch(xs, D, **kw)
    for b in range(xs.shape[0]):
        coef, r, info = refs[b]
        t, k, c = res.events[b]
        tag = (i, b, res.variant, xs.shape, D.shape)
        assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), tag
        assert np.array_equal(res.residuals[b], r), tag
        assert (res.coefficients[b] != coef).nnz == 0, tag
        assert res.stop_reasons()[b] == info['stop'], tag


N_EDGE = int(os.environ.get("HSCMP_FUZZ_EDGE", "120"))


@pytest.mark.parametrize('i', range(N_EDGE))
def test_medium_signals_with_atoms_piled_at_the_edges(i):
    """MFMA-sized problems (W up to 64, K up to 256, T a few thousand) whose energy sits at the two borders, where the
    zero- vs reflect-padding history of the rows matters (edge masks, stale reflected sample): against the oracle."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    rs = np.random.RandomState(41000 + i)
    dtype = np.float32 if i % 2 == 0 else np.float64
    W = int(rs.choice([8, 16, 32, 33, 64, 96, 127, 128])); K = int(rs.choice([7, 8, 32, 96, 100, 256]))
    T = int(rs.randint(3 * W, 20 * W + 200))
    D = rs.standard_normal((K, W)).astype(dtype)
    D /= np.sqrt(np.sum(np.square(D), axis=1, keepdims=True))
    x = (0.02 * rs.standard_normal(T)).astype(dtype)
    for _ in range(int(rs.randint(20, 60))):
        k = rs.randint(0, K); c = rs.uniform(0.5, 2.0) * rs.choice([-1.0, 1.0])
        side = rs.randint(0, 3)
        p = [rs.randint(-W // 2, W + 2), rs.randint(T - 2 * W - 2, T + W // 2), rs.randint(0, T)][side]   # centre, may stick out
        lo = p - (W - 1) // 2
        s, e = max(0, lo), min(T, lo + W)
        if e > s:
            x[s:e] += (c * D[k][s - lo:e - lo]).astype(dtype)
    kw = dict(nbNonzeroCoefs=int(rs.randint(30, 160)))
    if rs.rand() < 0.5:
        kw['nbBlocks'] = 'auto' if rs.rand() < 0.5 else int(rs.randint(2, 12))
    if rs.rand() < 0.3:
        kw['toleranceSnr'] = float(rs.uniform(10, 30))
    coef, res, info = orc.cmp_encode(x, D, maxEvents=1 << 16, **kw)
    if info['stop'] == 'capacity' or not np.all(np.isfinite(res)):
        return
    cmp = ConvolutionalMatchingPursuit()
    coefficients, residual = cmp.computeCoefficients(x, D, **kw)
    t, k, c = cmp.lastResult.events[0]
    tag = (i, cmp.lastResult.variant, T, K, W, kw)
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), tag
    assert np.array_equal(residual, res), tag
    assert cmp.lastResult.stop_reasons()[0] == info['stop'], tag


N_LEVEL = int(os.environ.get("HSCMP_FUZZ_LEVEL", "24"))


@pytest.mark.parametrize('i', range(N_LEVEL))
def test_level_shaped_medium_problems(i):
    """Level >= 1 shaped problems at realistic widths (W up to 65, up to 130 features): sparse inputs with a few dense
    rows, sparse dictionaries with singletons -- bucketed pair sort, LDS list capacities, row-list overflow, dictionary
    lists too long for LDS."""
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    rs = np.random.RandomState(51000 + i)
    W = int(rs.choice([9, 16, 33, 65])); F = int(rs.choice([12, 48, 130])); K = int(rs.randint(4, 24))
    T = int(rs.randint(4 * W, 7 * W + 80))                      # (sized for the oracle, which correlates densely)
    D = np.zeros((K, W, F))
    for k in range(K):
        for _ in range(int(rs.randint(1, 6))):
            D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
        D[k] /= np.sqrt(np.sum(np.square(D[k])))
    S = np.zeros((F, W, F)); S[np.arange(F), (W - 1) // 2, np.arange(F)] = 1.0
    D = np.concatenate((S, D), axis=0)
    x = np.zeros((T, F))
    for _ in range(int(rs.randint(T // 20, T // 3))):
        x[rs.randint(0, T), rs.randint(0, F)] = rs.uniform(0.3, 2.0) * rs.choice([-1.0, 1.0])
    for t in rs.randint(0, T, size=int(rs.randint(0, 4))):
        n = min(F, int(rs.randint(9, 30)))
        x[t, rs.permutation(F)[:n]] = rs.standard_normal(n)            # a dense-ish row: its feature list overflows
    w = np.ones(D.shape[0]); w[:F] = rs.uniform(0.7, 0.95)
    kw = dict(toleranceSnr=float(rs.uniform(10, 30)), nbNonzeroCoefs=150, nbBlocks=[1, 4, 10, 'auto'][i % 4], weights=w, minCoefficients=None)
    coef, res, info = orc.cmp_encode(x, D, maxEvents=600, **kw)     # (the dense oracle is slow here: bounded trace)
    if info['stop'] == 'capacity' or not np.all(np.isfinite(res)):
        return
    cmp = ConvolutionalMatchingPursuit()
    coefficients, residual = cmp.computeCoefficients(x, D, **kw)
    t, k, c = cmp.lastResult.events[0]
    tag = (i, cmp.lastResult.variant, T, K, W, F, len(info['t']))
    assert np.array_equal(t, info['t']) and np.array_equal(k, info['k']) and np.array_equal(c, info['c']), tag
    assert np.array_equal(residual, res), tag
    assert cmp.lastResult.stop_reasons()[0] == info['stop'], tag
