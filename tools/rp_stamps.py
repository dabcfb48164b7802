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
names = ['top of round', 'P1 rest (wait for the other waves)', 'P2/P3 rest (barrier, deferred slot stores)', '-', 'P4 subtraction', 'P5 re-correlation',
         'P6 rest (barriers, round end)', '  wave 0: filters + order', '  wave 0: prefix', '  wave 0: block arg-max', '  wave 0: candidate (k, c, energies)',
         '  wave 0: slot lookup', '  wave 0: segment scans']
for base, what in ((16, 'matrix-core policy (level 0)'), (0, 'sparse policy (levels >= 1)')):
    rounds, atoms = max(v[base + 14], 1), max(v[base + 15], 1)
    print('round-parallel loop, %s, workgroup 0: %d rounds, %d atoms (%.1f per round)' % (what, rounds, atoms, atoms / rounds))
    tot = 0.0
    for i, nm in enumerate(names):
        if v[base + i] > 0:
            print('  %-44s %9.0f cycles/round' % (nm, v[base + i] / rounds))
            tot += v[base + i]
    print('  %-44s %9.0f cycles/round' % ('total', tot / rounds))
cnt = (ctypes.c_ulonglong * 16)()
lib.hscmp_debug_counters(cnt, 1)
c = np.array(list(cnt), dtype=np.float64)
e, r = max(c[0], 1), max(c[3], 1)
print('sparse policy, workgroup 0 (both timed repetitions): %d energy calls, %d by the dense walk (%.2f %%), %.1f span cells on average' % (c[0], c[1], 100 * c[1] / e, c[2] / e))
print('  %d row-range attempts: %d with an overflowed row list, %d with too many cells, %d with too many products; %.1f cells, %.1f products, longest feature list %.1f on average; %d rows by their atoms' % (
    c[3], c[4], c[5], c[7], c[6] / r, c[8] / r, c[10] / r, c[9]))
nm = {32: 'energies: row lists + (k, c)', 33: 'energies: cells -> list', 34: 'energies: atom cells', 35: 'energies: ranks + partial sums', 36: 'energies: dense walk (if any) + tree',
      40: 're-correlation: gather', 41: 're-correlation: pairing', 42: 're-correlation: sort', 43: 're-correlation: chains', 44: 're-correlation: per-row best + stores'}
rounds = max(v[14], 1)
for i in sorted(nm):
    print('  wave 0, %-44s %9.0f cycles/round' % (nm[i], v[i] / rounds))
