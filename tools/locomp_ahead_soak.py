"""Soak of LoCOMP's side-by-side path (csrc/hscmp_locomp.h, locomp_precompute): the fuzz draws of tests/test_gpu_fuzz.py::_draw with
the signal repeated six times under varying gains and a blocked selection of 4 .. 7 blocks, so that most rounds have their selections more
than 5W + 8 samples apart; HSCMP_LOCOMP_AHEAD=7 (everything on) against =0, bit for bit (coefficients, residual, events, counters).
usage: python tools/locomp_ahead_soak.py FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, logging
logging.disable(logging.WARNING)
import test_gpu_fuzz as f
from hsc_amd.modeling import LoCOMP
from hsc_amd._native import HscmpError
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = ran = spaced = 0
for i in range(lo, hi):
    x, D, kw = f._draw(i); kw = dict(kw)
    if x.shape[0] > 800 or x.shape[0] < 2 * D.shape[1]: continue
    rs = np.random.RandomState(i)
    gains = rs.uniform(0.5, 1.5, size=6)
    x = np.concatenate([g * x for g in gains], axis=0).astype(x.dtype)
    kw['nbBlocks'] = int(rs.randint(4, 8))
    kw['nbNonzeroCoefs'] = min(6 * kw.get('nbNonzeroCoefs', 20), 240)
    if 'toleranceResidualScale' in kw: kw['toleranceResidualScale'] = float(kw['toleranceResidualScale'])
    out = {}
    try:
        for mode in ('7', '0'):
            os.environ['HSCMP_LOCOMP_AHEAD'] = mode
            c = LoCOMP(); res = c.computeCoefficientsBatch(np.stack([x, x[::-1].copy()]), D, **kw)
            out[mode] = res
    except HscmpError as ex:
        print(i, 'device error', str(ex)[:80]); continue
    a, s = out['7'], out['0']
    ran += 1
    same = np.array_equal(a.stats, s.stats) and np.array_equal(a.residuals, s.residuals) and all((a.coefficients[b] != s.coefficients[b]).nnz == 0 for b in range(2)) \
        and all(all(np.array_equal(u, v) for u, v in zip(a.events[b], s.events[b])) for b in range(2))
    if not same:
        bad += 1
        print(i, a.variant, x.shape, D.shape, {k: v for k, v in kw.items() if k != 'weights'}, 'MISMATCH', flush=True)
print('cases', ran, 'mismatching', bad)
