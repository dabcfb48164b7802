"""N > 1 path on CPU: two gloo ranks shard a batch, encode their shards (with the CPU oracle standing
in for the GPU engine -- test infrastructure), gather the per-signal results and must reproduce the
single-process result signal for signal."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_the_batch():
    from hsc_amd.parallel import shard_bounds
    for n in (1, 7, 8, 1024, 1031):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _OracleBatch(object):
    def __init__(self, sequences, D, **kw):
        from oracle import hsc_oracle as orc
        self.coefficients, res, self.events, stats, en = [], [], [], [], []
        for x in sequences:
            c, r, info = orc.cmp_encode(x, D, **kw)
            self.coefficients.append(c); res.append(r)
            self.events.append((info['t'], info['k'], info['c']))
            stats.append([info['nnz'], info['duplicates'], info['rounds'], 0, info['iterations'], len(info['t']), 0, 0])
            en.append([info['energy_signal'], info['energy_residual']])
        self.residuals = np.stack(res); self.stats = np.array(stats, dtype=np.int32); self.energies = np.array(en)


def _oracle_encode(sequences, D, **kw):
    return _OracleBatch(sequences, D, **kw)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import hsc_amd.synth as synth
    from hsc_amd.parallel import shard_bounds, encode_sharded, broadcast_dictionary
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        D = synth.make_dictionary(16, 16, seed=5) if rank == 0 else None
        D, _ = broadcast_dictionary(D, None, src=0)
        first, last = shard_bounds(7, world, rank)
        # every rank generates ITS OWN shard from the per-signal streams
        shard = synth.make_batch(D, 512, first, last - first, kind='planted', nb_atoms=12, seed=5)
        out = encode_sharded(shard, D, encode_fn=_oracle_encode, residuals=True, nbNonzeroCoefs=12)
        lean = encode_sharded(shard, D, encode_fn=_oracle_encode, nbNonzeroCoefs=12)       # default: per-signal results only
        assert lean['residuals'] is None and lean['bytes_per_signal'] == out['bytes_per_signal']
        if rank == 0:
            import golden_util as gu
            trip = [gu.csc_triplets(c) for c in out['coefficients']]
            np.savez(os.path.join(out_dir, 'gathered.npz'),
                     t=np.concatenate([e[0] for e in out['events']]), k=np.concatenate([e[1] for e in out['events']]),
                     c=np.concatenate([e[2] for e in out['events']]), n=np.array([len(e[0]) for e in out['events']]),
                     residuals=out['residuals'], stats=out['stats'], energies=out['energies'],
                     bytes_per_signal=out['bytes_per_signal'],
                     coef_row=np.concatenate([a[0] for a in trip]), coef_col=np.concatenate([a[1] for a in trip]),
                     coef_data=np.concatenate([a[2] for a in trip]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), 'gathered.npz'))
    import hsc_amd.synth as synth
    D = synth.make_dictionary(16, 16, seed=5)
    full = synth.make_batch(D, 512, 0, 7, kind='planted', nb_atoms=12, seed=5)
    ref = _oracle_encode(full, D, nbNonzeroCoefs=12)
    assert np.array_equal(got['n'], [len(e[0]) for e in ref.events])
    assert np.array_equal(got['t'], np.concatenate([e[0] for e in ref.events]))
    assert np.array_equal(got['k'], np.concatenate([e[1] for e in ref.events]))
    assert np.array_equal(got['c'], np.concatenate([e[2] for e in ref.events]))
    assert np.array_equal(got['residuals'], ref.residuals)
    assert got['stats'].shape == (7, 8) and np.array_equal(got['stats'][:, :5], ref.stats[:, :5])
    assert np.array_equal(got['energies'], ref.energies)
    # the payload of the collectives is the per-signal results only: a few KB per signal (north_star)
    assert 0 < int(got['bytes_per_signal']) <= 4096
    # the coefficient matrices rebuilt from the gathered events equal the encoder's own
    import golden_util as gu
    trip = [gu.csc_triplets(c) for c in ref.coefficients]
    assert np.array_equal(got['coef_row'], np.concatenate([a[0] for a in trip]))
    assert np.array_equal(got['coef_col'], np.concatenate([a[1] for a in trip]))
    assert np.array_equal(got['coef_data'], np.concatenate([a[2] for a in trip]))


def _bcast_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from hsc_amd.parallel import broadcast_dictionary
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(3)
        D = rs.standard_normal((6, 5, 3)) if rank == 0 else None            # float64, 3-D, with weights
        w = rs.uniform(0.5, 1.0, size=6) if rank == 0 else None
        D2, w2 = broadcast_dictionary(D, w, src=0)
        np.savez(os.path.join(out_dir, 'bcast%d.npz' % rank), D=D2, w=w2)
    finally:
        dist.destroy_process_group()


def test_dictionary_broadcast_is_a_tensor_broadcast(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_bcast_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(os.path.join(str(tmp_path), 'bcast%d.npz' % r)) for r in (0, 1))
    assert a['D'].dtype == np.float64 and a['D'].shape == (6, 5, 3)
    assert np.array_equal(a['D'], b['D']) and np.array_equal(a['w'], b['w'])


@pytest.mark.gpu
def test_gather_from_device_buffers_under_nccl():
    """The `nccl` leg of hsc_amd.parallel on the GPU box's one GPU (a one-rank process group, in a child process): the
    gather runs on device tensors that are views of the engine's buffers, the broadcast on a device tensor."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nccl_child.py')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, child, str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'NCCL-CHILD-OK' in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])


# ---- the hierarchical encoder across ranks (BASELINE configs[4]: "3-level HSC ... batch sharded 8 GPU") -------------------------

def _oracle_level_coder(D):
    from oracle import hsc_oracle as orc

    class Coder(object):
        def encode(self, X, **kw):
            c, r, _ = orc.cmp_encode(np.asarray(X), D, **kw)
            return c, r
    return Coder()


def _hier_inputs():
    """The small 3-level dictionary of tests/golden/hsc_small.npz and seven related signals."""
    import golden_util as gu
    from hsc_amd.dataset import MultilevelDictionary
    z = gu.load('hsc_small.npz')
    mld = MultilevelDictionary.fromRawDictionaries([z['raw0'], z['raw1'], z['raw2']], [int(s) for s in z['scales']]).withSingletonBases()
    rs = np.random.RandomState(12)
    x = z['x']
    xs = np.stack([x, 0.5 * x, x[::-1].copy(), np.roll(x, 17), x + 0.02 * rs.standard_normal(x.shape).astype(x.dtype),
                   np.roll(x[::-1], 5).copy(), 2.0 * x]).astype(x.dtype)
    return mld, xs, dict(toleranceSnr=[15.0, 20.0, 25.0], nbBlocks=4, singletonWeight=0.9)


def _oracle_hier_encode(xs, mld, **kw):
    """The host logic of the hierarchical encoder (hsc/modeling.py:1427-1654) on the CPU oracle as level coder: the stand-in for
    the GPU batch entry in the CPU tests -- (per-signal level matrices, residual energies, event records)."""
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    from hsc_amd.dataset import convertSparseMatricesToEvents
    h = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    h._level_coder = _oracle_level_coder
    coefs, energies, events = [], [], []
    for x in xs:
        c, r = h.computeCoefficients(x, mld, **kw)
        coefs.append(c); energies.append(float(np.sum(np.square(np.asarray(r, dtype=np.float64)))))
        events.append(convertSparseMatricesToEvents(c))
    return coefs, np.array(energies), events


def _hier_worker(rank, world, port, out_dir, nsig=7):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import pickle
    import torch.distributed as dist
    from hsc_amd.parallel import shard_bounds, encode_sharded_hierarchical
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mld, xs, kw = _hier_inputs()
        xs = xs[:nsig]
        first, last = shard_bounds(xs.shape[0], world, rank)
        out = encode_sharded_hierarchical(xs[first:last], mld, encode_fn=_oracle_hier_encode, **kw)
        lean = encode_sharded_hierarchical(xs[first:last], mld, encode_fn=_oracle_hier_encode, exact=False, **kw)
        assert lean['bytes_total'] < out['bytes_total']
        if rank == world - 1:                       # (every rank holds every signal's results: check the last one, not rank 0)
            with open(os.path.join(out_dir, 'hier.pkl'), 'wb') as f:
                pickle.dump(dict(out=out, lean_coefficients=lean['coefficients']), f)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_hierarchical_gather(tmp_path):
    """Two gloo ranks shard seven signals, run the 3-level encode on their shards and gather events (the reference's wire format,
    hsc/dataset.py:798-811), float64 values, counts and residual energies: every signal's per-level matrices, events and energy equal
    the single-process result."""
    import pickle
    import socket
    import golden_util as gu
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_hier_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    with open(os.path.join(str(tmp_path), 'hier.pkl'), 'rb') as f:
        got = pickle.load(f)
    mld, xs, kw = _hier_inputs()
    coefs, energies, events = _oracle_hier_encode(xs, mld, **kw)
    out = got['out']
    assert len(out['coefficients']) == xs.shape[0] and np.array_equal(out['energies'], energies)
    for b in range(xs.shape[0]):
        assert np.array_equal(out['events'][b], events[b]), b
        for l in range(3):
            a, e = gu.csc_triplets(out['coefficients'][b][l]), gu.csc_triplets(coefs[b][l])
            assert out['coefficients'][b][l].shape == coefs[b][l].shape
            assert all(np.array_equal(u, v) for u, v in zip(a, e)), (b, l)           # float64 values travelled: bit for bit
            f32 = gu.csc_triplets(got['lean_coefficients'][b][l])
            assert np.array_equal(f32[0], e[0]) and np.array_equal(f32[2], e[2].astype(np.float32).astype(np.float64)), (b, l)
    # the payload is per-signal results only: 24 bytes per coefficient (16-byte record + float64 value) + 16 bytes
    nev = max(sum(len(events[b]) for b in range(0, 4)), sum(len(events[b]) for b in range(4, 7)))
    assert 0 < out['bytes_per_signal'] <= 16384 and out['bytes_total'] == 24 * nev + 16 * 4


def test_three_rank_gloo_hierarchical_gather_with_an_empty_shard(tmp_path):
    """Two signals over three ranks: shard sizes 1, 1, 0 -- the rank without a signal still takes part in every collective and ends
    up with both signals' results."""
    import pickle
    import socket
    import golden_util as gu
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_hier_worker, args=(3, port, str(tmp_path), 2), nprocs=3, join=True)
    with open(os.path.join(str(tmp_path), 'hier.pkl'), 'rb') as f:
        got = pickle.load(f)['out']
    mld, xs, kw = _hier_inputs()
    coefs, energies, events = _oracle_hier_encode(xs[:2], mld, **kw)
    assert len(got['coefficients']) == 2 and np.array_equal(got['energies'], energies)
    for b in range(2):
        assert np.array_equal(got['events'][b], events[b])
        for l in range(3):
            assert all(np.array_equal(u, v) for u, v in zip(gu.csc_triplets(got['coefficients'][b][l]), gu.csc_triplets(coefs[b][l])))
