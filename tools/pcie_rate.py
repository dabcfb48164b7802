"""PCIe-inclusive rate of the hot path: host buffers in (hscmp_encode_batch), results out to host.
Never the headline `value` of bench.py (that one has inputs resident in HBM); noted in DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsc_amd.synth as synth
from hsc_amd import _native
B = int(os.environ.get('B', '1024'))
D = synth.make_dictionary(256, 64, seed=2)
x = np.tile(synth.make_batch(D, 65536, 0, 16, kind='planted', nb_atoms=256, seed=2), (B // 16, 1))[:, :, np.newaxis]
eng = _native.Engine(0); eng.set_dictionary(D)
params = _native.make_params(nbNonzeroCoefs=256, eps=float(np.finfo(np.float32).eps), maxEvents=576)
for rep in range(3):
    t0 = time.perf_counter()
    eng.encode_batch(x, params)                       # H2D + kernels, synchronous
    t1 = time.perf_counter()
    stats = eng.fetch_stats(); ev = eng.fetch_events(); sl = eng.fetch_slots(); en = eng.fetch_energies()
    t2 = time.perf_counter()
    r = eng.fetch_residual()
    t3 = time.perf_counter()
    n = int(stats[:, _native.STAT_ITERATIONS].sum())
    print('rep %d: encode_batch (H2D %.0f MB + kernels) %.1f ms; fetch events/slots/stats %.1f ms; fetch residual (%.0f MB) %.1f ms; '
          'selections/s: %.2fM (events only)  %.2fM (with residual)'
          % (rep, x.nbytes / 1e6, 1e3 * (t1 - t0), 1e3 * (t2 - t1), r.nbytes / 1e6, 1e3 * (t3 - t2), n / (t2 - t0) / 1e6, n / (t3 - t0) / 1e6), flush=True)
