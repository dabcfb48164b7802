"""Diagnostic: per-phase cycle shares of the LoCOMP atom body (csrc/hscmp_locomp.h) at one level of a BASELINE hierarchical
configuration (libhscmp built with -DHSCMP_DBG_STAMPS, path in argv[1]); env CONFIG (4 / 5, default 4), LEVEL (default 0),
B (signals, default 256), PACK (signals per workgroup of the matrix-core policy)."""
import ctypes, logging, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hsc_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
import bench_hsc
from hsc_amd.modeling import LoCOMP
logging.disable(logging.WARNING)
B = int(os.environ.get('B', '256')); CONFIG = int(os.environ.get('CONFIG', '4')); LEVEL = int(os.environ.get('LEVEL', '0'))
if os.environ.get('PACK'):
    os.environ['HSCMP_LOCOMP_PACK'] = os.environ['PACK']
mlds, xs, kw, desc = bench_hsc.build_workload(CONFIG, B, 65536, 0, 17)
lib = _native.load_library()
out = (ctypes.c_ulonglong * 64)()
inp = xs
for level in range(LEVEL + 1):
    D = mlds.getRawDictionary(level)
    nbS = D.shape[0] - mlds.countsNoSingletons[level]
    w = np.ones((D.shape[0],), dtype=D.dtype); w[:nbS] = kw.get('singletonWeight', 0.5)
    lib.hscmp_debug_stamps(out, 1)
    res = LoCOMP().computeCoefficientsBatch(inp, D, toleranceSnr=kw['toleranceSnr'][level], nbBlocks=kw['nbBlocks'], weights=w)
    if level < LEVEL:
        inp = np.stack([c.toarray() for c in res.coefficients], axis=0)
lib.hscmp_debug_stamps(out, 1)
v = np.array(list(out), dtype=np.float64)
print('config', CONFIG, 'level', LEVEL, D.shape, res.variant, 'loop %.1f ms' % res.kernel_ms[2], 'selections per signal %.0f' % res.stats[:, 4].mean())
full, atoms = max(v[14], 1), v[15]
sel = max(float(res.stats[0, 4]), 1.0)
print('workgroup 0, signal 0: %d selections, %d of them through the whole atom body (the others: computed ahead and committed by their wave), their group size %.2f on average' % (sel, full, atoms / full))
print('  group sizes of those (bins of 8 atoms, last: >= 120):', ' '.join('%d' % v[16 + i] for i in range(16)))
print('  all stamp slots (cycles / selection):', {i: int(v[i] / sel) for i in range(64) if v[i] > 0 and i not in (14, 15) and not (16 <= i < 32)})
names = {33: 'ahead: scan', 34: 'ahead: order', 35: 'ahead: normal equations', 36: 'ahead: solve', 37: 'ahead: cells / subtractions', 38: 'ahead: wait for the slowest wave', 58: 'round: lookups + spacing', 59: 'round: ahead (all of it)', 60: 'round: atom (all of it)', 62: 'round: rows that waited (behind the last batch)', 63: 'ahead: rows that waited', 0: 'neighbourhood scan', 1: 'group order', 2: 'right-hand sides + Gram entries', 3: 'Cholesky + substitutions', 4: 'coefficients, subtractions, energies',
         5: 're-correlation (all of it)', 6: 'segments marked', 10: '  window', 11: '  tiles'}
tot = 0.0
for i in sorted(names):
    if v[i] > 0:
        print('  %-40s %8.0f cycles / selection' % (names[i], v[i] / sel))
    if i < 10:
        tot += v[i]
print('  %-40s %8.0f cycles / selection (the atom body; the kernel spends %.0f per selection at 2.4 GHz)' % ('total', tot / sel, 1e-3 * res.kernel_ms[2] * 2.4e9 / max(res.stats[0, 4], 1)))
