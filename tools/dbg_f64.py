import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, golden_util as gu
from hsc_amd.modeling import ConvolutionalMatchingPursuit
from oracle import hsc_oracle as orc
for name in ['f64_T300_K24_W10_snr12_nb2', 'f64_T300_K24_W10_snr12_nb4', 'f64_T128_K32_W9_snr20']:
    x, D, kw, exp = gu.small_case(name)
    cmp = ConvolutionalMatchingPursuit(); cmp.computeCoefficients(x, D, **kw)
    t, k, c = cmp.lastResult.events[0]
    _, _, info = orc.cmp_encode(x, D, **kw)
    n = min(len(t), len(info['t']))
    bad = np.where((t[:n] != info['t'][:n]) | (k[:n] != info['k'][:n]) | (c[:n] != info['c'][:n]))[0]
    print(name, cmp.lastResult.variant, 'n', len(t), len(info['t']), 'first bad', bad[:5], 'rounds', cmp.lastResult.stats[0][2], info['rounds'])
    if len(bad):
        i = bad[0]; print('   gpu', t[i], k[i], repr(c[i]), ' orc', info['t'][i], info['k'][i], repr(info['c'][i]))
