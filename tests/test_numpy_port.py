"""The NumPy port timed as bench.py's cpu_baseline must itself reproduce the real reference
(golden vectors), otherwise its timing would be of a different algorithm."""
import time

import numpy as np
import pytest

import golden_util as gu
from oracle import numpy_port as port

def _single_argmax(name):
    # the port covers the benchmark's path: one arg-max per round, no atom weights
    kw = gu.small_case(name)[2]
    return kw.get('nbBlocks', 1) == 1 and 'weights' not in kw


CASES = [n for n in gu.small_case_names() if n != 'f32_planted_T256_K4_W32' and _single_argmax(n)]


@pytest.mark.parametrize('name', CASES)
def test_port_matches_reference(name):
    x, D, kw, exp = gu.small_case(name)
    coefficients, residual, (t, k, c) = port.cmp_encode(x, D, **kw)
    assert np.array_equal(t, exp['t']) and np.array_equal(k, exp['k'])
    tol = 1e-5 if np.result_type(x.dtype, D.dtype) == np.float32 else 1e-10
    assert gu.rel_err(c, exp['c']) <= tol
    row, col, data = gu.csc_triplets(coefficients)
    assert np.array_equal(row, exp['row']) and np.array_equal(col, exp['col'])
    assert residual.shape == exp['residual'].shape and residual.dtype == exp['residual'].dtype


def test_port_config1_golden():
    import hsc_amd.synth as synth
    z = gu.load('cmp_config.npz')
    D = synth.make_dictionary(32, 32, seed=1)
    x = synth.make_signal(D, 4096, 0, kind='planted', nb_atoms=64, seed=1)
    _, _, (t, k, c) = port.cmp_encode(x, D, nbNonzeroCoefs=64)
    assert np.array_equal(t, z['config1_planted__t']) and np.array_equal(k, z['config1_planted__k'])
    assert gu.rel_err(c, z['config1_planted__c']) <= 1e-5
