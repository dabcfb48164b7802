"""Batch sharding of the matching-pursuit path over the GPUs of one node.

The reference has no distributed runtime; the signals of a batch are
independent (no cross-signal term in hsc/modeling.py:1053-1186), so the path shards by contiguous
blocks of signals, one process per GPU, with NO collective on the data path.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests) is used only to
broadcast the dictionary when the caller asks for it and to gather the per-signal results:
fixed-shape tensors -- events [b, n] (position, atom, coefficient in selection order), counters
[b, 8], energies [b, 2] -- through all_gather_into_tensor, about 3 KB per signal at BASELINE
config 2.  Residuals ([T] samples per signal) travel only on request.
"""
import numpy as np


def shard_bounds(n_signals, world_size, rank):
    """Contiguous block [first, last) of signals owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(int(n_signals), int(world_size))
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def _dist():
    import torch.distributed as dist
    return dist


FORCE_COLLECTIVES = False     # run the collectives even in a one-rank group (tests/nccl_child.py: RCCL on the box's one GPU)


def _comm_device(device=None):
    """Tensors of a collective live on the GPU with nccl (RCCL moves device memory), on the host with gloo.  Without an
    explicit device a rank uses cuda:LOCAL_RANK -- the rule the engine follows -- and makes it current: a process that
    never called torch.cuda.set_device would otherwise put every rank's tensors on cuda:0 (duplicate-GPU error or a hang)."""
    import os
    import torch
    dist = _dist()
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        if device is None:
            device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count()))
        torch.cuda.set_device(device)
        return device
    return torch.device('cpu')


def _collectives_on():
    dist = _dist()
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def broadcast_dictionary(D, weights=None, src=0, device=None):
    """Every rank returns rank `src`'s (D, weights): one small header broadcast (shape / dtype), then the
    dictionary itself as a tensor -- 64 KB at config 2, 6.3 MB for the largest level dictionary of config 4."""
    import torch
    dist = _dist()
    if not _collectives_on():
        return D, weights
    dev = _comm_device(device)
    me = dist.get_rank()
    hdr = torch.zeros(8, dtype=torch.int64, device=dev)
    if me == src:
        D = np.ascontiguousarray(D)
        hdr[0] = D.ndim
        hdr[1:1 + D.ndim] = torch.tensor(D.shape, dtype=torch.int64)
        hdr[5] = 1 if D.dtype == np.float64 else 0
        hdr[6] = 0 if weights is None else 1
    dist.broadcast(hdr, src=src)
    h = hdr.cpu().numpy()
    shape = tuple(int(v) for v in h[1:1 + int(h[0])])
    npdt = np.float64 if int(h[5]) else np.float32
    tdt = torch.float64 if int(h[5]) else torch.float32
    buf = torch.from_numpy(np.ascontiguousarray(D, dtype=npdt)).to(dev) if me == src else torch.empty(shape, dtype=tdt, device=dev)
    dist.broadcast(buf, src=src)
    Dout = buf.cpu().numpy()
    wout = None
    if int(h[6]):
        wb = torch.from_numpy(np.ascontiguousarray(weights, dtype=npdt)).to(dev) if me == src else torch.empty((shape[0],), dtype=tdt, device=dev)
        dist.broadcast(wb, src=src)
        wout = wb.cpu().numpy()
    return Dout, wout


class _DevicePointer(object):
    """A raw device allocation of the engine as a torch tensor (no copy), through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(int(s) for s in shape), typestr=typestr, data=(int(ptr), False), version=2)


def _engine_result_tensors(eng, device):
    """(stats [b,8] int32, energies [b,2] float64, ev_t, ev_k [b,cap] int32, ev_c [b,cap] dtype) of the engine's last
    encode as tensors on `device`: views of the engine's own GPU buffers when the collective runs on the GPU
    (nothing crosses PCIe), host fetches otherwise."""
    import torch
    B, _, cap = eng._batch
    if device.type == 'cuda':
        strict = _dist().is_initialized() and _dist().get_backend() == 'nccl'
        try:
            v = eng.device_view()
            ct = '<f8' if eng.dtype == np.float64 else '<f4'
            stats = torch.as_tensor(_DevicePointer(v.stats, (B, 8), '<i4'), device=device)
            ev_t = torch.as_tensor(_DevicePointer(v.ev_t, (B, cap), '<i4'), device=device)
            ev_k = torch.as_tensor(_DevicePointer(v.ev_k, (B, cap), '<i4'), device=device)
            ev_c = torch.as_tensor(_DevicePointer(v.ev_c, (B, cap), ct), device=device)
            en = torch.as_tensor(_DevicePointer(v.energies, (B, 2), ct), device=device).to(torch.float64)
            eng.synchronize()
            return stats, en, ev_t, ev_k, ev_c
        except Exception:
            if strict:
                raise                              # under RCCL a broken view must not hide behind host fetches
            # (no array-interface import in this torch build: go through the host)
    t, k, c = eng.fetch_events()
    return (torch.from_numpy(eng.fetch_stats()).to(device), torch.from_numpy(eng.fetch_energies()).to(device),
            torch.from_numpy(t).to(device), torch.from_numpy(k).to(device), torch.from_numpy(c).to(device))


def gather_results(source, device=None, residuals=None):
    """All ranks' per-signal results, in rank order, on every rank.

    source : an hsc_amd._native.Engine holding this rank's last encode, or a dict with numpy / torch arrays
             'stats' [b,8] int32, 'energies' [b,2], 'ev_t', 'ev_k' [b,cap] int32, 'ev_c' [b,cap]
    residuals : optional [b,T(,F)] array of this rank, gathered as a tensor too (opt-in: T samples per signal)
    Returns numpy arrays: 'stats' [Btot,8], 'energies' [Btot,2], 'ev_t', 'ev_k', 'ev_c' [Btot,n] (n = the longest
    event list of any rank; counts in stats[:,5]), 'bytes_per_signal' (payload of the collectives), 'residuals'.
    """
    import torch
    dist = _dist()
    dev = _comm_device(device)
    if isinstance(source, dict):
        as_t = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev)
        stats, en, ev_t, ev_k, ev_c = (as_t(source[k]) for k in ('stats', 'energies', 'ev_t', 'ev_k', 'ev_c'))
        en = en.to(torch.float64)
    else:
        stats, en, ev_t, ev_k, ev_c = _engine_result_tensors(source, dev)
    world = dist.get_world_size() if dist.is_initialized() else 1
    collect = _collectives_on()
    b = int(stats.shape[0])
    nmax = int(stats[:, 5].max().item()) if b else 0                    # HSCMP_STAT_EVENTS
    # ranks may own different numbers of signals / event counts: agree on the padded shape (two scalars)
    meta = torch.tensor([b, nmax], dtype=torch.int64, device=dev)
    if collect:
        metas = torch.empty((world * 2,), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(metas, meta)
        metas = metas.reshape((world, 2))
    else:
        metas = meta[None]
    metas = metas.cpu().numpy()
    bpad, n = int(metas[:, 0].max()), int(metas[:, 1].max())

    def padded(x, cols=None):
        x = x if cols is None else x[:, :cols]
        if cols is not None and x.shape[1] < cols:
            x = torch.cat([x, torch.zeros((x.shape[0], cols - x.shape[1]), dtype=x.dtype, device=dev)], dim=1)
        if x.shape[0] < bpad:
            x = torch.cat([x, torch.zeros((bpad - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)], dim=0)
        return x.contiguous()

    def gather(x):
        if not collect:
            return x.cpu().numpy()
        out = torch.empty((world * bpad,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
        dist.all_gather_into_tensor(out, x)
        out = out.cpu().numpy().reshape((world, bpad) + tuple(x.shape[1:]))
        return np.concatenate([out[r, :int(metas[r, 0])] for r in range(world)], axis=0)

    parts = dict(stats=padded(stats), energies=padded(en), ev_t=padded(ev_t, n), ev_k=padded(ev_k, n), ev_c=padded(ev_c, n))
    payload = sum(int(np.prod(v.shape[1:])) * v.element_size() for v in parts.values())
    out = {k: gather(v) for k, v in parts.items()}
    out['bytes_per_signal'] = payload
    out['residuals'] = None
    if residuals is not None:
        r = residuals if torch.is_tensor(residuals) else torch.from_numpy(np.ascontiguousarray(residuals))
        out['residuals'] = gather(padded(r.to(dev)))
    return out


def events_to_coefficients(ev_t, ev_k, ev_c, n, shape, minCoefficients=1e-16):
    """The reference's coefficient matrix from an ordered event list (hsc/modeling.py:1114 `+=` in selection order,
    float64; epilogue :1171-1181): duplicates accumulate in event order, exactly as the engine's slots do."""
    import scipy.sparse
    t = np.asarray(ev_t[:n], dtype=np.int64); k = np.asarray(ev_k[:n], dtype=np.int64)
    key = t * shape[1] + k
    uniq, inv = np.unique(key, return_inverse=True)
    acc = np.zeros(len(uniq), dtype=np.float64)
    np.add.at(acc, inv, np.asarray(ev_c[:n], dtype=np.float64))         # unbuffered: sequential in event order
    keep = acc != 0.0
    if minCoefficients is not None:
        keep &= np.abs(acc) >= minCoefficients
    m = scipy.sparse.coo_matrix((acc[keep], (uniq[keep] // shape[1], uniq[keep] % shape[1])), shape=shape).tocsc()
    return m


def encode_sharded(sequences_shard, D, encode_fn=None, gather=True, residuals=False, minCoefficients=1e-16, **kwargs):
    """Encode this rank's shard and (optionally) gather every rank's per-signal results.

    sequences_shard : [b_local, T] or [b_local, T, F] -- the signals this rank owns
    encode_fn       : callable(sequences, D, **kwargs) -> object with .events, .stats, .energies, .residuals (and
                      .coefficients); defaults to the GPU engine (ConvolutionalMatchingPursuit.computeCoefficientsBatch
                      on cuda:LOCAL_RANK)
    residuals       : also gather the residuals (off by default: T samples per signal against ~3 KB of results)
    Returns a dict; with gather=True every rank gets the results of ALL signals in rank order:
      'events'   list of (t, k, c) per signal (selection order)      'stats' int32 [B_total, 8]
      'energies' float64 [B_total, 2]     'coefficients' list of csc_matrix (rebuilt from the events)
      'residuals' [B_total, T(,F)] or None     'bytes_per_signal' gathered payload per signal
    """
    dist = _dist()
    if encode_fn is None:
        import os
        from .modeling import ConvolutionalMatchingPursuit
        cmp = ConvolutionalMatchingPursuit(device=int(os.environ.get('LOCAL_RANK', '0')))
        encode_fn = cmp.computeCoefficientsBatch
        kwargs = dict(kwargs, minCoefficients=minCoefficients)
    res = encode_fn(sequences_shard, D, **kwargs)
    T = sequences_shard.shape[1]
    K = D.shape[0]
    if not gather or not _collectives_on():
        return dict(events=res.events, stats=np.asarray(res.stats), energies=np.asarray(res.energies),
                    coefficients=res.coefficients, residuals=np.asarray(res.residuals) if residuals else None, bytes_per_signal=0)
    b = len(res.events)
    cap = max([len(e[0]) for e in res.events] + [1])
    cdt = res.events[0][2].dtype if b else np.float32
    ev_t = np.zeros((b, cap), dtype=np.int32); ev_k = np.zeros((b, cap), dtype=np.int32); ev_c = np.zeros((b, cap), dtype=cdt)
    stats = np.array(res.stats, dtype=np.int32, copy=True).reshape((b, 8))
    for i, (t, k, c) in enumerate(res.events):
        ev_t[i, :len(t)] = t; ev_k[i, :len(t)] = k; ev_c[i, :len(t)] = c
        stats[i, 5] = len(t)
    g = gather_results(dict(stats=stats, energies=np.asarray(res.energies, dtype=np.float64), ev_t=ev_t, ev_k=ev_k, ev_c=ev_c),
                       residuals=np.asarray(res.residuals) if residuals else None)
    out = dict(stats=g['stats'], energies=g['energies'], residuals=g['residuals'], bytes_per_signal=g['bytes_per_signal'], events=[], coefficients=[])
    for i in range(g['stats'].shape[0]):
        n = int(g['stats'][i, 5])
        out['events'].append((g['ev_t'][i, :n].copy(), g['ev_k'][i, :n].copy(), g['ev_c'][i, :n].copy()))
        out['coefficients'].append(events_to_coefficients(g['ev_t'][i], g['ev_k'][i], g['ev_c'][i], n, (T, K), minCoefficients))
    return out
