"""Diagnostic: per-phase cycle shares of the LoCOMP atom body (csrc/hscmp_locomp.h) at level 0 of BASELINE config 4
(libhscmp built with -DHSCMP_DBG_STAMPS, path in argv[1]); env B (signals, default 256), PACK (signals per workgroup)."""
import ctypes, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hsc_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
import bench_hsc
from hsc_amd.modeling import LoCOMP
B = int(os.environ.get('B', '256'))
if os.environ.get('PACK'):
    os.environ['HSCMP_LOCOMP_PACK'] = os.environ['PACK']
mlds, xs, kw, desc = bench_hsc.build_workload(4, B, 65536, 0, 17)
D0 = mlds.getRawDictionary(0)
lib = _native.load_library()
out = (ctypes.c_ulonglong * 64)()
coder = LoCOMP()
for rep in range(2):
    lib.hscmp_debug_stamps(out, 1)
    res = coder.computeCoefficientsBatch(xs, D0, toleranceSnr=kw['toleranceSnr'][0], nbBlocks=kw['nbBlocks'])
lib.hscmp_debug_stamps(out, 1)
v = np.array(list(out), dtype=np.float64)
print(res.variant, 'loop %.1f ms' % res.kernel_ms[2], 'selections per signal %.0f' % res.stats[:, 4].mean())
sel, atoms = max(v[14], 1), v[15]
print('workgroup 0, signal 0: %d selections, group size %.2f on average' % (sel, atoms / sel))
names = {0: 'neighbourhood scan', 1: 'group order', 2: 'right-hand sides + Gram entries', 3: 'Cholesky + substitutions', 4: 'coefficients, subtractions, energies',
         5: 're-correlation (all of it)', 6: 'segments marked', 10: '  window', 11: '  tiles', 12: '  rows resolved'}
tot = 0.0
for i in sorted(names):
    print('  %-40s %8.0f cycles / selection' % (names[i], v[i] / sel))
    if i < 10:
        tot += v[i]
print('  %-40s %8.0f cycles / selection (the atom body; the kernel spends %.0f per selection)' % ('total', tot / sel, 1e-3 * res.kernel_ms[2] * 2.4e9 / res.stats[0, 4]))
