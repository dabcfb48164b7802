"""Diagnostic (GPU box): invariants of the greedy loop checked inside the kernel, build -DHSCMP_DBG_CHECKSEG.

    hipcc <flags of csrc/Makefile> -DHSCMP_DBG_CHECKSEG -shared -o tools/dbg/libhscmp_CHECKSEG.so hierarchical-sparse-coding_amd/csrc/hscmp_api.hip
    python tools/check_invariants.py tools/dbg/libhscmp_CHECKSEG.so

  * at the start of every blocked selection round: the segment maxima kept in LDS == a fresh scan of the score array;
  * after every atom applied near a signal end: each re-correlated edge row == a fresh pinned chain over the final
    residual (this is the check that caught the reflected-sample load/store race).
Runs the config-4 level-0 workload (640 signals of 8192 samples, blocked selection) with one signal per workgroup and
with four, and compares the event lists of the two."""
import ctypes
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hsc_amd.synth as synth
from hsc_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
from hsc_amd.modeling import ConvolutionalMatchingPursuit

B, T = 640, 8192
mld = synth.make_hierarchy(W1=16, seed=4)
xs = synth.make_hierarchy_batch(mld, T, 0, B, seed=4)
D = mld.getRawDictionary(0)
lib = _native.load_library()
out = (ctypes.c_ulonglong * 16)()
f32 = lambda u: struct.unpack('f', struct.pack('I', u & 0xffffffff))[0]
ref = None
for q in ('0', '1'):
    os.environ['HSCMP_MFMA_QUAD'] = q
    for r in range(int(os.environ.get('NRUNS', '6'))):
        lib.hscmp_debug_counters(out, 1)
        res = ConvolutionalMatchingPursuit().computeCoefficientsBatch(xs, D, toleranceSnr=30.0, nbBlocks=10)
        lib.hscmp_debug_counters(out, 0)
        v = list(out)
        if ref is None:
            ref = res
        nbad = sum(1 for b in range(B) if len(ref.events[b][0]) != len(res.events[b][0]) or any(
            not np.array_equal(ref.events[b][i], res.events[b][i]) for i in range(3)))
        print('signals per workgroup %s, run %d: segment-maxima mismatches %d, edge-row mismatches %d, signals differing from run 0: %d'
              % ('4' if q == '1' else '1', r, v[0], v[1], nbad))
        for n in range(min(3, v[1])):
            a, bb, c, d = v[4 + 4 * n:8 + 4 * n]
            print('   signal %d row t=%d: fresh %.9g kept %.9g; atom p=%d; first differing tap %d holds %.9g'
                  % (a >> 32, a & 0x7fffffff, f32(bb >> 32), f32(bb), c >> 32, d >> 48, f32(d)))
