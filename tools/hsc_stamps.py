"""Diagnostic: per-phase cycle shares of the generic (non-fused) atom body on a level-1 shaped input
(libhscmp built with -DHSCMP_DBG_STAMPS, path in argv[1])."""
import ctypes, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsc_amd.synth as synth
from hsc_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
from hsc_amd.dataset import MultilevelDictionary
from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit

B = int(os.environ.get('B', '8')); T = int(os.environ.get('T', '65536'))
K0, W0, K1, W1 = 256, 64, 128, 16
CONFIG5 = os.environ.get('CONFIG') == '5'
rs = np.random.RandomState(11)
D0 = synth.make_dictionary(K0, W0, seed=4)
D1 = np.zeros((K1, W1, K0), dtype=np.float32)
for k in range(K1):
    for _ in range(3):
        D1[k, rs.randint(0, W1), rs.randint(0, K0)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
    D1[k] /= np.sqrt(np.sum(D1[k] ** 2))
mld = MultilevelDictionary.fromRawDictionaries([D0, D1], [W0, W0 + W1 - 1]).withSingletonBases()
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
snr, blocks = [30.0, 40.0], 10
if CONFIG5:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import generate_dataset as gd
    mld = gd.build([32, 64, 128], 4.0, patience=100)
    xs, _, _ = gd.signals(mld, B, T)
    mld = mld.withSingletonBases()
    snr = [30.0, 35.0, 35.0]
lib = _native.load_library()
out = (ctypes.c_ulonglong * 64)()
hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp')
lib.hscmp_debug_stamps(out, 1)
coefs, residuals, timings = hcmp.computeCoefficientsBatch(xs, mld, toleranceSnr=snr, nbBlocks=blocks, singletonWeight=0.95)
lib.hscmp_debug_stamps(out, 1)
v = np.array(list(out), dtype=np.float64)
n = max(v[46], 1)
for tm in timings:
    print('level %d: %-28s init %.2f ms  loop %.2f ms  selections %d' % (tm['level'], tm['variant'], tm['kernel_ms'][1], tm['kernel_ms'][2], tm['selections']))
nf = max(v[15], 1)
print('level 0 (fused body), workgroup 0: %d atoms' % nf)
for i, nm in enumerate(['loads + resolve + update', 'B1', 'energy', 'MFMA tile', 'B4', 'segments + bookkeeping', 'B5', 'deferred stores']):
    print('  %-28s %9.0f cycles/atom' % (nm, v[i] / nf))
print('  total %.0f cycles/atom (the selection between atoms is not stamped)' % (v[:8].sum() / nf))
names = ['select (per round)', 'bookkeeping', 'residual update', 're-correlation', 'segments + stop', '-', '-', 'slow stop rules (per round)']
print('workgroup 0: %d atoms' % n)
for i, nm in enumerate(names):
    print('  %-28s %9.0f cycles/atom' % (nm, v[32 + i] / n))
print('  total %.0f cycles/atom' % ((v[32:40].sum() + v[48:50].sum()) / n))
print('    select / block arg-max %9.0f, compactions + sort %9.0f, weak-atom filter + rest = the select line above' % (v[48] / n, v[49] / n))
sub = ['gather', 'pairing', 'sort', 'chains', 'row arg-max', 'list append (before the gather)']
for i, nm in enumerate(sub):
    print('    re-correlation / %-14s %9.0f cycles/atom' % (nm, v[40 + i] / n))
print('    re-correlation / row arg-max: %.0f cycles/atom of it in thread 0\'s own rows, the rest at the barrier' % (v[51] / n))
print('    residual update (merged form) / first barrier %.0f, gather loop %.0f, barrier %.0f, atom cells + barrier %.0f, partial-sum registration + barrier %.0f, stores + sums %.0f cycles/atom' % tuple(v[52:58] / n))
cnt = (ctypes.c_ulonglong * 16)()
lib.hscmp_debug_counters(cnt, 1)
c = np.array(list(cnt), dtype=np.float64)
calls = max(c[0], 1)
print('  sparse_rows calls %d (all levels >= 1 of workgroup 0, initial correlation included): window non-zeros %.1f avg (%d overflows), occupied rows %.1f, pairs %.1f avg (%d overflows)' % (
    calls, c[1] / calls, c[2], c[5] / calls, c[3] / calls, c[4]))
