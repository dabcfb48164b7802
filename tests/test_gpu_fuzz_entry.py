"""Seeded random sweep (-m gpu) of the row-level entry points against the CPU oracle, bit for bit:
convolve1d ('same' / 'valid'), _selectBestAtoms (single / blocked / 'auto', offsets, weights, null threshold),
_updateInnerProducts (reflect padding at both edges, multi-bounce for short signals), window assignment of the
k-means learner."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = int(os.environ.get('HSCMP_FUZZ_ENTRY', '160'))


def _shape(i):
    rs = np.random.RandomState(31000 + i)
    dtype = np.float32 if rs.rand() < 0.5 else np.float64
    W = int(rs.randint(1, 40)); K = int(rs.randint(1, 40)); F = int(rs.choice([1, 1, 2, 5]))
    T = int(rs.randint(W, 12 * W + 30))
    D = rs.standard_normal((K, W, F)).astype(dtype)
    x = rs.standard_normal((T, F)).astype(dtype)
    return rs, dtype, T, K, W, F, x, D


@pytest.mark.parametrize('i', range(N))
def test_convolve1d_random(i):
    from hsc_amd.modeling import convolve1d
    from oracle import hsc_oracle as orc
    rs, dtype, T, K, W, F, x, D = _shape(i)
    xs, Ds = (x[:, 0], D[:, :, 0]) if F == 1 and rs.rand() < 0.5 else (x, D)
    for padding in ('same', 'valid'):
        got = convolve1d(xs, Ds, padding=padding)
        exp = orc.convolve1d(xs, Ds, padding=padding)
        assert got.shape == exp.shape and got.dtype == exp.dtype and np.array_equal(got, exp), (i, padding, T, K, W, F)


@pytest.mark.parametrize('i', range(N))
def test_select_best_atoms_random(i):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    from oracle import hsc_oracle as orc
    rs, dtype, T, K, W, F, x, D = _shape(i)
    ip = rs.standard_normal((T, K)).astype(dtype)
    ip[rs.rand(T, K) < 0.2] = 0.0
    if rs.rand() < 0.3:
        ip[rs.randint(0, T, size=3)] = ip[rs.randint(0, T)]          # equal rows: ties between positions
    cmp = ConvolutionalMatchingPursuit()
    modes = [1, 'auto'] + ([int(rs.randint(2, max(3, T // 2)))] if T >= 6 else [])
    for nb in modes:
        for offset in (False, True):
            w = rs.uniform(0.3, 1.0, size=K).astype(dtype) if rs.rand() < 0.5 else None
            thres = float(rs.choice([0.0, 1e-16, 0.3]))
            try:
                t, k, c = orc.select_best_atoms(ip, W, nbBlocks=nb, offset=offset, nullCoeffThres=thres, weights=w)
            except Exception:
                continue                                                  # (block size 0 etc.: an error in the reference too)
            atoms = cmp._selectBestAtoms(ip, W, nbBlocks=nb, offset=offset, nullCoeffThres=thres, weights=w)
            tag = (i, nb, offset, thres, T, K, W)
            assert [a.position for a in atoms] == t.tolist() and [a.index for a in atoms] == k.tolist(), tag
            assert np.array_equal(np.array([a.coefficient for a in atoms], dtype=dtype), c), tag


@pytest.mark.parametrize('i', range(N))
def test_update_inner_products_random(i):
    from hsc_amd.modeling import ConvolutionalMatchingPursuit, Atom
    from oracle import hsc_oracle as orc
    rs, dtype, T, K, W, F, x, D = _shape(i)
    ip = orc.convolve1d(x, D, padding='same')
    ipo = ip.copy()
    cmp = ConvolutionalMatchingPursuit()
    for _ in range(4):
        p = int(rs.choice([0, T - 1, rs.randint(0, T), min(T - 1, W), max(0, T - 1 - W)]))
        r = rs.standard_normal((T, F)).astype(dtype)
        cmp._updateInnerProducts(ip, r, [Atom(p, 0, 1.0, W)], D)
        orc.update_inner_products(ipo, r, D, p)
        assert np.array_equal(ip, ipo), (i, p, T, K, W, F)


@pytest.mark.parametrize('i', range(N))
def test_assign_windows_random(i):
    from hsc_amd import _native
    from oracle import hsc_oracle as orc
    rs, dtype, T, K, W, F, x, D = _shape(i)
    L = int(rs.randint(W, 3 * W + 2))
    n = int(rs.randint(1, 12))
    windows = rs.standard_normal((n, L, F)).astype(dtype)
    windows[rs.rand(n, L, F) < 0.3] = 0.0
    eng = _native.Engine(0)
    eng.set_dictionary(D)
    t, k, c = eng.assign_windows(windows)
    for q in range(n):
        ipq = orc.convolve1d(windows[q], D, padding='valid')
        o = int(np.argmax(np.abs(ipq).reshape(-1)))
        assert (t[q], k[q]) == (o // K, o % K) and c[q] == ipq[o // K, o % K], (i, q, L, K, W, F)
    eng.close()
