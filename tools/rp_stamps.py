"""Diagnostic: per-phase cycle shares of the round-parallel loop (csrc/hscmp_rp.h) on the BASELINE config-5 workload
(libhscmp built with -DHSCMP_DBG_STAMPS, path in argv[1]); env B (signals, default 128), LEVELS (how many levels to run)."""
import ctypes, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hsc_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
import bench_hsc
from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit

B = int(os.environ.get('B', '128')); T = int(os.environ.get('T', '65536')); CONFIG = int(os.environ.get('CONFIG', '5'))
mlds, xs, kw, desc = bench_hsc.build_workload(CONFIG, B, T, 0)
lib = _native.load_library()
out = (ctypes.c_ulonglong * 64)()
hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp')
for rep in range(2):
    lib.hscmp_debug_stamps(out, 1)
    coefs, energies, timings = hcmp.computeCoefficientsBatch(xs, mlds, residuals='energy', **kw)
lib.hscmp_debug_stamps(out, 1)
v = np.array(list(out), dtype=np.float64)
for tm in timings:
    print('level %d: %-34s prepare %.2f init %.2f loop %.2f ms  selections %d' % (tm['level'], tm['variant'], tm['kernel_ms'][0], tm['kernel_ms'][1], tm['kernel_ms'][2], tm['selections']))
rounds, atoms = max(v[30], 1), max(v[31], 1)
print('round-parallel loops of workgroup 0 (all levels that ran it): %d rounds, %d atoms (%.1f per round)' % (rounds, atoms, atoms / rounds))
names = {16: 'top of round', 17: 'P1 rest (wait for the other waves)', 18: 'P2/P3 rest (barrier)', 19: 'P3 prefix', 20: 'P4 bookkeeping stores + subtraction',
         21: 'P5 re-correlation', 22: 'P6 rest (barriers, round end)', 23: '  wave 0: filters + order', 24: '  wave 0: prefix', 25: '  wave 0: block arg-max', 26: '  wave 0: candidate (resolve + energies)', 27: '  wave 0: slot lookup', 28: '  wave 0: segment scans', 29: '-'}
tot = 0.0
for i in range(16, 30):
    if v[i] > 0:
        print('  %-40s %9.0f cycles/round' % (names[i], v[i] / rounds))
        tot += v[i]
print('  %-40s %9.0f cycles/round = %.1f us at 2.4 GHz' % ('total', tot / rounds, tot / rounds / 2400.0))
