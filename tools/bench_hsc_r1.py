"""Hierarchical (2-level) encode at a scaled-down BASELINE config-4 shape: level 0 = K0 x W0 single
feature (MFMA path), level 1 = (K0 singletons + K1) x W1 x K0 on the level-0 coefficient streams
(float64, sparse multi-feature path).  Prints per-level kernel times; with HSCMP_FORCE_GENERIC=1 the
dense generic kernels run instead (same results) for comparison."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsc_amd.synth as synth
from hsc_amd.dataset import MultilevelDictionary
from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit

B = int(os.environ.get('B', '32')); T = int(os.environ.get('T', '8192'))
K0, W0, K1, W1 = int(os.environ.get('K0', '64')), int(os.environ.get('W0', '32')), int(os.environ.get('K1', '32')), int(os.environ.get('W1', '16'))
rs = np.random.RandomState(11)
D0 = synth.make_dictionary(K0, W0, seed=4)
D1 = np.zeros((K1, W1, K0), dtype=np.float32)
for k in range(K1):                                    # sparse composite atoms: 3 events each
    for _ in range(3):
        D1[k, rs.randint(0, W1), rs.randint(0, K0)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
    D1[k] /= np.sqrt(np.sum(D1[k] ** 2))
scales = [W0, W0 + W1 - 1]
mld = MultilevelDictionary.fromRawDictionaries([D0, D1], scales).withSingletonBases()
rep1 = mld.getMultiscaleDictionaries()[1]
xs = []
for b in range(B):
    x = 0.01 * rs.standard_normal(T)
    for _ in range(T // 128):
        i = rs.randint(K0, rep1.shape[0]); t = rs.randint(64, T - 64); c = rs.uniform(0.5, 2.0) * rs.choice([-1.0, 1.0])
        s, e, es, ee = synth.centered_span(T, rep1.shape[1], t)
        x[s:e] += c * rep1[i][es:ee]
    xs.append(x.astype(np.float32))
xs = np.stack(xs)
hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp')
for rep in range(2):
    t0 = time.perf_counter()
    coefs, residuals, timings = hcmp.computeCoefficientsBatch(xs, mld, toleranceSnr=[30.0, 40.0], nbBlocks=10, singletonWeight=0.95,
                                                             memoryBudget=(float(os.environ['MEM']) if 'MEM' in os.environ else None))
    wall = time.perf_counter() - t0
snr = 10 * np.log10(np.sum(xs.astype(np.float64) ** 2) / np.sum(residuals ** 2))
print('B=%d T=%d L0 %dx%d, L1 (%d+%d)x%dx%d  wall %.2f s  SNR %.1f dB  nnz/level %s' % (
    B, T, K0, W0, K0, K1, W1, K0, wall, snr, [int(np.mean([c[l].nnz for c in coefs])) for l in range(2)]))
for tm in timings:
    print('  level %d: %-28s init %.2f ms  loop %.2f ms  selections %d (%d re-selections)' % (tm['level'], tm['variant'], tm['kernel_ms'][1], tm['kernel_ms'][2], tm['selections'], tm.get('duplicates', 0)))
