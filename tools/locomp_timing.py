"""Diagnostic (GPU box): LoCOMP (device-resident table, host loop) beside the greedy coder, per signal.
Config-2 dictionary (256 atoms x 64 taps), T = 8192 and 65536, blocked selection, 20 dB."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hsc_amd.synth as synth
from hsc_amd.modeling import LoCOMP, ConvolutionalMatchingPursuit

D = synth.make_dictionary(256, 64, seed=1)
for T in (8192, 65536):
    x = synth.make_signal(D, T, 0, kind='planted', nb_atoms=T // 256, seed=1)
    for cls in (ConvolutionalMatchingPursuit, LoCOMP):
        m = cls()
        for rep in range(2):
            t0 = time.perf_counter()
            c, r = m.computeCoefficients(x, D, toleranceSnr=20.0, nbBlocks=10)
            dt = time.perf_counter() - t0
        print('%-30s T %6d nnz %5d  %.3f s  snr %.1f dB' % (cls.__name__, T, c.nnz, dt, 10 * np.log10(np.sum(x.astype(np.float64) ** 2) / np.sum(r.astype(np.float64) ** 2))), flush=True)
m = LoCOMP()
cProfile.run("m.computeCoefficients(x, D, toleranceSnr=20.0, nbBlocks=10)", '/tmp/lo.prof')
pstats.Stats('/tmp/lo.prof').sort_stats('cumulative').print_stats(14)
