"""Hierarchical (3-level) encode of generated data in the style of BASELINE config 5: scales [32, 64, 128]
(window widths [32, 33, 65]), K_l = 4 x width patterns per level (+ singleton bases), Poisson-event
signals (tools/generate_dataset.py).  Prints per-level kernel times of the device-chained batch encoder."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generate_dataset as gd
from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit

B = int(os.environ.get('B', '64')); T = int(os.environ.get('T', '65536'))
t0 = time.time()
mld = gd.build([32, 64, 128], float(os.environ.get('OC', '4')), patience=int(os.environ.get('PATIENCE', '100')))
t1 = time.time()
xs, evs, rates = gd.signals(mld, B, T, rate=float(os.environ.get('RATE', '5e-4')), compression=None if os.environ.get('NOSCALE') else 0.25)
mld = mld.withSingletonBases()
print('dictionary %.1f s, signals %.1f s; raw dictionaries %s; %d events per signal' % (
    t1 - t0, time.time() - t1, [d.shape for d in mld.dictionaries], np.mean([len(e) for e in evs])), flush=True)
hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp')
for rep in range(2):
    t0 = time.perf_counter()
    coefs, residuals, timings = hcmp.computeCoefficientsBatch(xs, mld, toleranceSnr=[30.0, 35.0, 35.0], nbBlocks=10, singletonWeight=0.95,
                                                             memoryBudget=(float(os.environ['MEM']) if 'MEM' in os.environ else None))
    wall = time.perf_counter() - t0
snr = 10 * np.log10(np.sum(xs.astype(np.float64) ** 2) / np.sum(residuals ** 2))
print('B=%d T=%d wall %.2f s  SNR %.1f dB  nnz/level %s' % (B, T, wall, snr, [int(np.mean([c[l].nnz for c in coefs])) for l in range(3)]))
for tm in timings:
    print('  level %d: %-32s init %.2f ms  loop %.2f ms  selections %d (%d re-selections)' % (tm['level'], tm['variant'], tm['kernel_ms'][1], tm['kernel_ms'][2], tm['selections'], tm.get('duplicates', 0)))
