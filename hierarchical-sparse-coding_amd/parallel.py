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
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:           # ('cuda' without an index: the current device)
            device = torch.device('cuda', torch.cuda.current_device())
        torch.cuda.set_device(device)       # (RCCL binds a communicator to the current device: it stays current afterwards)
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
    if device is None and not isinstance(source, dict) and getattr(source, 'device', None) is not None \
            and dist.is_initialized() and dist.get_backend() == 'nccl':
        device = torch.device('cuda', int(source.device))       # the engine's own GPU: its raw pointers are only valid there
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


# ---- the hierarchical encoder (hsc/modeling.py:1636-1643): gather of the per-signal multilevel results ------------------------
# What a rank holds per signal after HierarchicalConvolutionalMatchingPursuit.computeCoefficientsBatch(returnEvents=True): the event
# records of hsc/dataset.py:798-811 -- (time int32, level int32, index int32, coefficient float32), 16 bytes each, sorted by time --
# the float64 coefficient behind every record (the reference's matrices are float64, :1074; its wire format rounds to float32) and the
# residual energy.  They travel as ONE ragged buffer per rank (the records of its signals back to back, padded to the longest rank
# only) plus the per-signal counts: no [signals x longest list] padding.

def _pack_records(events):
    """Event records of this rank's signals -> (int32 [n, 4] with the float32 coefficient bit-cast, counts int64 [b])."""
    counts = np.array([len(e) for e in events], dtype=np.int64)
    rec = np.zeros((int(counts.sum()), 4), dtype=np.int32)
    o = 0
    for e in events:
        n = len(e)
        rec[o:o + n, 0] = e['f0']; rec[o:o + n, 1] = e['f1']; rec[o:o + n, 2] = e['f2']
        rec[o:o + n, 3] = np.ascontiguousarray(e['f3'], dtype=np.float32).view(np.int32)
        o += n
    return rec, counts


def _unpack_records(rec):
    from .dataset import EVENT_DTYPE
    e = np.zeros((rec.shape[0],), dtype=EVENT_DTYPE)
    e['f0'] = rec[:, 0]; e['f1'] = rec[:, 1]; e['f2'] = rec[:, 2]
    e['f3'] = np.ascontiguousarray(rec[:, 3]).view(np.float32)
    return e


def events_to_level_matrices(events, counts, T, values64=None):
    """Per-level csc_matrix [T, count] (float64) of one signal from its event records -- convertEventsToSparseMatrices
    (hsc/dataset.py:813-824) with the float64 values when they came along (`values64`), else the records' float32 ones."""
    import scipy.sparse
    v = np.asarray(events['f3'], dtype=np.float64) if values64 is None else np.asarray(values64, dtype=np.float64)
    t = np.asarray(events['f0'], dtype=np.int64); l = np.asarray(events['f1']); i = np.asarray(events['f2'], dtype=np.int64)
    out = []
    for level, count in enumerate(counts):
        m = l == level
        out.append(scipy.sparse.coo_matrix((v[m], (t[m], i[m])), shape=(int(T), int(count))).tocsc())
    return out


def gather_hierarchical(events, energies, values64=None, device=None):
    """All ranks' multilevel per-signal results, in rank order, on every rank.

    events   : list (this rank's signals) of event record arrays (hsc_amd.dataset.EVENT_DTYPE)
    energies : float64 [b] residual energies          values64 : optional list of float64 arrays, one value per record
    Returns {'events': list over ALL signals, 'values64': list or None, 'energies': float64 [B_total], 'counts': int64 [B_total],
             'bytes_per_signal': mean payload of the collectives per signal, 'bytes_total': payload of this rank}"""
    import torch
    dist = _dist()
    dev = _comm_device(device)
    rec, counts = _pack_records(events)
    en = np.ascontiguousarray(energies, dtype=np.float64).reshape((-1,))
    assert en.shape[0] == counts.shape[0]
    vals = None if values64 is None else (np.concatenate([np.asarray(v, dtype=np.float64) for v in values64]) if len(values64) else np.zeros((0,)))
    assert vals is None or vals.shape[0] == rec.shape[0]
    world = dist.get_world_size() if dist.is_initialized() else 1
    collect = _collectives_on()
    b, n = int(counts.shape[0]), int(rec.shape[0])
    meta = torch.tensor([b, n], dtype=torch.int64, device=dev)
    if collect:
        metas = torch.empty((world * 2,), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(metas, meta)
        metas = metas.reshape((world, 2)).cpu().numpy()
    else:
        metas = meta[None].cpu().numpy()
    bpad, npad = int(metas[:, 0].max()), int(metas[:, 1].max())

    def gather(a, pad, sizes):
        x = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        if x.shape[0] < pad:
            x = torch.cat([x, torch.zeros((pad - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)], dim=0)
        if not collect:
            return a, int(x.numel() * x.element_size())
        out = torch.empty((world * pad,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
        dist.all_gather_into_tensor(out, x.contiguous())
        out = out.cpu().numpy().reshape((world, pad) + tuple(x.shape[1:]))
        return np.concatenate([out[r, :int(sizes[r])] for r in range(world)], axis=0), int(x.numel() * x.element_size())

    payload = 0
    g_counts, nb = gather(counts, bpad, metas[:, 0]); payload += nb
    g_en, nb = gather(en, bpad, metas[:, 0]); payload += nb
    g_rec, nb = gather(rec, npad, metas[:, 1]); payload += nb
    g_vals = None
    if vals is not None:
        g_vals, nb = gather(vals, npad, metas[:, 1]); payload += nb
    ev_all = _unpack_records(g_rec)
    starts = np.concatenate(([0], np.cumsum(g_counts)))
    out_events = [ev_all[int(starts[i]):int(starts[i + 1])] for i in range(len(g_counts))]
    out_vals = None if g_vals is None else [g_vals[int(starts[i]):int(starts[i + 1])] for i in range(len(g_counts))]
    return dict(events=out_events, values64=out_vals, energies=g_en, counts=g_counts, bytes_total=payload,
                bytes_per_signal=payload / max(1, bpad))


def _values_of_events(coefficients, events):
    """The float64 value of every event record, from the per-level matrices the records were made of."""
    mats = [c.tocsr() for c in coefficients]
    out = np.zeros((len(events),), dtype=np.float64)
    l = np.asarray(events['f1'])
    for level, m in enumerate(mats):
        sel = np.where(l == level)[0]
        if len(sel):
            out[sel] = np.asarray(m[np.asarray(events['f0'])[sel], np.asarray(events['f2'])[sel]]).reshape((-1,))
    return out


def encode_sharded_hierarchical(sequences_shard, multilevelDict, encode_fn=None, gather=True, exact=True, method='cmp', **kwargs):
    """The hierarchical encoder on this rank's shard, then (optionally) every rank's per-signal multilevel results on every rank.

    encode_fn : callable(sequences, multilevelDict, **kwargs) -> (per-signal lists of per-level matrices, residual energies [b],
                per-signal event records); defaults to HierarchicalConvolutionalMatchingPursuit(method).computeCoefficientsBatch on
                cuda:LOCAL_RANK with the device epilogue (events and energies come out of hscmp_hierarchy_epilogue)
    exact     : the float64 coefficients travel beside the float32 records, so the rebuilt matrices equal the encoder's bit for bit
    Returns {'coefficients': per signal a list of per-level csc_matrix, 'events', 'energies', 'bytes_per_signal', 'bytes_total'}."""
    if encode_fn is None:
        import os
        from .hierarchical import HierarchicalConvolutionalMatchingPursuit
        hcmp = HierarchicalConvolutionalMatchingPursuit(method=method, device=int(os.environ.get('LOCAL_RANK', '0')))

        def encode_fn(xs, mld, **kw):
            coefs, energies, _, events = hcmp.computeCoefficientsBatch(xs, mld, returnEvents=True, residuals='energy', **kw)
            return coefs, energies, events
    coefs, energies, events = encode_fn(sequences_shard, multilevelDict, **kwargs)
    T = sequences_shard.shape[1]
    counts = [int(multilevelDict.getRawDictionary(l).shape[0]) for l in range(multilevelDict.getNbLevels())]
    if not gather or not _collectives_on():
        return dict(coefficients=coefs, events=events, energies=np.asarray(energies, dtype=np.float64), bytes_per_signal=0, bytes_total=0)
    vals = [_values_of_events(c, e) for c, e in zip(coefs, events)] if exact else None
    g = gather_hierarchical(events, energies, vals)
    mats = [events_to_level_matrices(e, counts, T, None if g['values64'] is None else g['values64'][i]) for i, e in enumerate(g['events'])]
    return dict(coefficients=mats, events=g['events'], energies=g['energies'], bytes_per_signal=g['bytes_per_signal'], bytes_total=g['bytes_total'])
