"""Diagnostic: loop throughput vs workgroups per CU (HSCMP_LDS_PAD forces lower occupancy)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hsc_amd.synth as synth
from hsc_amd import _native
K = int(os.environ.get('K', '128')); B = int(os.environ.get('B', '768')); L0 = 128
D = synth.make_dictionary(K, 64, seed=2)
x = torch.from_numpy(synth.make_batch(D, 65536, 0, 8, kind='planted', nb_atoms=L0, seed=2)).cuda().repeat(B // 8, 1).contiguous()
eng = _native.Engine(0); eng.set_dictionary(D)
params = _native.make_params(nbNonzeroCoefs=L0, eps=1.2e-7, maxEvents=2 * L0 + 64)
ms = []
for i in range(4):
    eng.encode_batch_device(x.data_ptr(), B, 65536, params); eng.synchronize(); ms.append(eng.last_kernel_ms().copy())
ms = np.array(ms)[1:].mean(0)
nsel = int(eng.fetch_stats()[:, 4].sum())
print('K=%d B=%d pad=%s: loop %.3f ms, %.2f M selections/s in the loop, loop MFMA %.1f TF' % (
    K, B, os.environ.get('HSCMP_LDS_PAD', '0'), ms[2], nsel / ms[2] / 1e3, 2.0 * 127 * K * 64 * nsel / ms[2] / 1e9))
