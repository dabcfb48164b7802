"""Child process of tests/test_parallel.py::test_gather_from_device_buffers_under_nccl: a one-rank `nccl` (= RCCL) process
group on the one GPU of the box, with the collectives forced on (parallel.FORCE_COLLECTIVES): `broadcast` and
`all_gather_into_tensor` really execute in RCCL.  The collectives of hsc_amd.parallel then run on device tensors that are views of the
engine's own buffers (no host round trip) -- the path `bench.py --gpus N` takes on a multi-GPU node."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsc_amd.synth as synth
from hsc_amd import _native, parallel

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % sys.argv[1], world_size=1, rank=0)
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
parallel.FORCE_COLLECTIVES = True      # a one-rank group: broadcast and all_gather_into_tensor still go through RCCL
D = synth.make_dictionary(32, 16, seed=3)
x = synth.make_batch(D, 2048, 0, 12, kind='planted', nb_atoms=20, seed=3)
Db, wb = parallel.broadcast_dictionary(D, None, src=0, device=dev)
assert np.array_equal(Db, D) and wb is None
eng = _native.Engine(0)
eng.set_dictionary(D)
eng.encode_batch(x[:, :, None], _native.make_params(nbNonzeroCoefs=20, eps=1.2e-7, maxEvents=64))
g = parallel.gather_results(eng, device=dev)
t, k, c = eng.fetch_events()
stats = eng.fetch_stats()
n = int(stats[:, 5].max())
assert np.array_equal(g['stats'], stats) and np.array_equal(g['energies'], eng.fetch_energies().astype(np.float64))
assert np.array_equal(g['ev_t'], t[:, :n]) and np.array_equal(g['ev_k'], k[:, :n]) and np.array_equal(g['ev_c'], c[:, :n])
assert g['bytes_per_signal'] <= 4096
r = eng.fetch_residual()
g2 = parallel.gather_results(eng, device=dev, residuals=r)
assert np.array_equal(g2['residuals'], r)
# with no device named, the collectives run on cuda:LOCAL_RANK (the engine's rule), not on whatever device is current
g3 = parallel.gather_results(eng)
assert np.array_equal(g3['ev_t'], g['ev_t'])
# the device views really are views: no copy was made of the event buffers
st, en, et, ek, ec = parallel._engine_result_tensors(eng, dev)
assert et.is_cuda and et.data_ptr() == eng.device_view().ev_t
# the hierarchical encoder's gather (BASELINE configs[4]): the real GPU batch entry with the device epilogue, its event records,
# float64 values, counts and residual energies through RCCL; the rebuilt per-level matrices equal the encoder's own
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_parallel import _hier_inputs
from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
mld, xs, kw = _hier_inputs()
hc = HierarchicalConvolutionalMatchingPursuit(method='cmp', device=0)
coefs, energies, _, events = hc.computeCoefficientsBatch(xs, mld, returnEvents=True, residuals='energy', **kw)
out = parallel.encode_sharded_hierarchical(xs, mld, method='cmp', **kw)
assert out['bytes_total'] > 0 and np.array_equal(out['energies'], energies)
for b in range(xs.shape[0]):
    assert np.array_equal(out['events'][b], events[b])
    for l in range(mld.getNbLevels()):
        assert (out['coefficients'][b][l] != coefs[b][l]).nnz == 0 and out['coefficients'][b][l].nnz == coefs[b][l].nnz
hc.close()
dist.destroy_process_group()
print('NCCL-CHILD-OK')
