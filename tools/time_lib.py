"""Diagnostic (GPU box): kernel times of the config-2 workload for one or more builds of libhscmp.so (ablation builds).
usage: time_lib.py <libhscmp*.so> [...]     (each library is timed in a child process)"""
import os
import subprocess
import sys

if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] == '--all'):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), lib], check=False)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hsc_amd.synth as synth
from hsc_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
B = int(os.environ.get('B', '1024'))
D = synth.make_dictionary(256, 64, seed=2)
x = torch.from_numpy(synth.make_batch(D, 65536, 0, 8, kind='planted', nb_atoms=256, seed=2)).cuda().repeat(B // 8, 1).contiguous()
eng = _native.Engine(0)
eng.set_dictionary(D)
params = _native.make_params(nbNonzeroCoefs=256, eps=1.2e-7, maxEvents=576)
ms = []
for i in range(6):
    eng.encode_batch_device(x.data_ptr(), B, 65536, params)
    eng.synchronize()
    ms.append(eng.last_kernel_ms())
best = min(ms[2:], key=lambda m: m[2])
print('%-44s prepare %.3f  init %.3f  loop %.3f ms' % (os.path.basename(sys.argv[1]), best[0], best[1], best[2]), flush=True)
