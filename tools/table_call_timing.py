"""Diagnostic (GPU box): per-call times of the device-resident LoCOMP table entry points (hscmp_table_open / _select) at the\nconfig-2, config-4 level-1 and long-signal shapes (sparse multi-feature input at level 1)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hsc_amd import _native
rs = np.random.RandomState(0)
for (T, K, W, F, dt) in ((8192, 256, 64, 1, np.float32), (8192, 384, 17, 256, np.float64), (65536, 256, 64, 1, np.float32)):
    D = rs.standard_normal((K, W, F)).astype(dt); D /= np.sqrt((D**2).sum(axis=(1,2), keepdims=True))
    eng = _native.Engine(0); eng.set_dictionary(D)
    x = rs.standard_normal((T, F)).astype(dt)
    if F > 1: x = x * (rs.rand(T, F) < 0.002)
    tab = eng.table_open(x); eng.synchronize()
    t0 = time.perf_counter(); tab = eng.table_open(x); eng.synchronize(); t1 = time.perf_counter()
    for nb in (1, 10):
        eng.table_select(nb, False, 1e-16, None)
        t2 = time.perf_counter()
        for _ in range(20): eng.table_select(nb, False, 1e-16, None)
        t3 = time.perf_counter()
        print('T %d K %d W %d F %d %s: open %.2f ms, select(nbBlocks=%d) %.3f ms/call' % (T, K, W, F, np.dtype(dt).name, 1e3*(t1-t0), nb, 1e3*(t3-t2)/20), flush=True)
    eng.close()
