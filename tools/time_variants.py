"""Diagnostic: time corr_init / iterate of alternative builds of libhscmp (ablation studies)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hsc_amd.synth as synth
from hsc_amd import _native
B = int(os.environ.get('B', '1024'))
DT = np.float64 if os.environ.get('DTYPE', 'f32') == 'f64' else np.float32
D = synth.make_dictionary(256, 64, seed=2, dtype=DT)
x = torch.from_numpy(synth.make_batch(D, 65536, 0, 8, kind='planted', nb_atoms=256, seed=2, dtype=DT)).cuda().repeat(B // 8, 1).contiguous()
for lib in sys.argv[1:]:
    _native._lib = None; _native._engines.clear()
    _native.LIB_PATH = os.path.abspath(lib)
    eng = _native.Engine(0); eng.set_dictionary(D)
    params = _native.make_params(nbNonzeroCoefs=256, eps=float(np.finfo(DT).eps), maxEvents=576)
    ms = []
    for i in range(4):
        eng.encode_batch_device(x.data_ptr(), B, 65536, params); eng.synchronize(); ms.append(eng.last_kernel_ms().copy())
    ms = np.array(ms)[1:].mean(0)
    print('%-40s %-34s init %.3f ms (%.1f TF)  loop %.3f ms' % (os.path.basename(lib), eng.last_variant(), ms[1], 2.199023 * B / 1024 / ms[1] * 1e3, ms[2]), flush=True)
