"""Soak of the LoCOMP device loop (csrc/hscmp_locomp.h) against the host loop (np.linalg.pinv per group, as the reference) on the
random configurations of tests/test_gpu_fuzz.py: support, coefficients, residual.  What it reports as mismatching on a healthy build
are groups of (nearly) dependent atoms -- tiny signals under large dictionaries, a composite atom next to the singletons it is made
of: the two solvers then pick different least-squares solutions with the same residual, or an entry cancels to
exactly 0.0 in one of them.      usage: python tools/locomp_soak.py FIRST_SEED LAST_SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_gpu_fuzz as f
from hsc_amd.modeling import LoCOMP
from hsc_amd._native import HscmpError
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
import logging; logging.disable(logging.WARNING)
for i in range(lo, hi):
    x, D, kw = f._draw(i)
    kw = dict(kw)
    if kw.get('nbNonzeroCoefs', 0) and kw['nbNonzeroCoefs'] > 40: kw['nbNonzeroCoefs'] = 40
    if x.shape[0] > 1500: continue
    os.environ.pop('HSCMP_LOCOMP_HOST', None)
    dev = LoCOMP()
    try:
        cd, rd = dev.computeCoefficients(x, D, **kw)
        reason = dev.lastResult.stop_reasons()[0]; variant = dev.lastResult.variant
    except HscmpError as ex:
        print(i, 'device error', str(ex)[:80]); continue
    os.environ['HSCMP_LOCOMP_HOST'] = '1'
    try:
        ch, rh = LoCOMP().computeCoefficients(x, D, **kw)
    except Exception as ex:
        print(i, 'host error', type(ex).__name__, str(ex)[:80]); continue
    a, h = cd.tocsc(), ch.tocsc()
    tol = 1e-4 if x.dtype == np.float32 or D.dtype == np.float32 else 1e-8
    scale = max(1.0, abs(h).max() if h.nnz else 1.0)
    same_support = a.nnz == h.nnz and np.array_equal(a.indices, h.indices) and np.array_equal(a.indptr, h.indptr)
    dc = abs(a - h).max() if (a.nnz or h.nnz) else 0.0
    dr = float(np.max(np.abs(rd.astype(np.float64) - rh.astype(np.float64))))
    ok = same_support and dc <= tol * scale
    if not ok:
        bad += 1
        print(i, variant, reason, x.shape, D.shape, {k: v for k, v in kw.items() if k != 'weights'}, 'nnz', a.nnz, h.nnz, 'support', same_support, 'dc %.2e dr %.2e' % (dc, dr), flush=True)
print('cases', hi - lo, 'mismatching', bad)
