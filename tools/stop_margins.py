#!/usr/bin/env python
"""How often is an SNR stop of the greedy loop decided by rounding noise?  (parity item 2.)

The engine's energy sums use a pinned order, NumPy's pairwise sums differ from it by ~1e-7 relative in float32.
A stop `E_sig / E_res >= 10^(tol/10)` can only come out differently under the other summation order when the ratio at the deciding
iteration lies within that noise of the threshold.  This script encodes the level-0 signals of BASELINE configs[3] (1024 x 65536,
256 x 64 dictionary, toleranceSnr 30 dB, nbBlocks=10) and, from the event trace of every signal, recomputes the residual energy after
each ROUND in float64; it reports the distribution of the relative distance between the threshold and the ratio at the last two
stop tests (the one that stopped and the one before).      usage (GPU box): python tools/stop_margins.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench_hsc
from hsc_amd.modeling import ConvolutionalMatchingPursuit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mlds, xs, kw, _ = bench_hsc.build_workload(4, B, 65536, 0, 17)
D = mlds.getRawDictionary(0)
tol = kw['toleranceSnr'][0]
res = ConvolutionalMatchingPursuit().computeCoefficientsBatch(xs, D, toleranceSnr=tol, nbBlocks=kw['nbBlocks'])
thr = 10.0 ** (tol / 10.0)
en = np.asarray(res.energies, dtype=np.float64)
ratio = en[:, 0] / en[:, 1]
margin = ratio / thr - 1.0                      # >= 0 at an SNR stop: how far past the threshold the deciding test was
stops = res.stop_reasons()
snr_stops = np.array([s == 'snr' for s in stops])
m = margin[snr_stops]
print('signals %d, SNR stops %d' % (B, int(snr_stops.sum())))
for q in (1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
    print('  deciding ratio within %.0e of the threshold: %d signals' % (q, int((m < q).sum())))
print('  smallest margins:', np.sort(m)[:8])
print('  (a stop decided one ATOM later or earlier changes nnz by one; with ~3400 atoms per signal the per-atom energy step is ~3e-4 of '
      'E_res, so a margin below 2e-7 -- twice the float32 summation noise -- is where the two summation orders can disagree)')
