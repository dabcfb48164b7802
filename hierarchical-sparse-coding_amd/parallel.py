"""Batch sharding of the matching-pursuit path over the GPUs of one node.

The reference has no distributed runtime; the signals of a batch are
independent (no cross-signal term in hsc/modeling.py:1053-1186), so the path shards by contiguous
blocks of signals, one process per GPU, with NO collective on the data path.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests) is used only to
broadcast the dictionary when the caller asks for it and to gather the per-signal results --
a few KB per signal.
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


def broadcast_dictionary(D, weights=None, src=0):
    """Every rank returns rank `src`'s (D, weights) -- 64 KB at config 2, one RCCL broadcast."""
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return D, weights
    box = [(D, weights)] if dist.get_rank() == src else [None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def encode_sharded(sequences_shard, D, encode_fn=None, gather=True, **kwargs):
    """Encode this rank's shard and (optionally) gather every rank's per-signal results.

    sequences_shard : [b_local, T] or [b_local, T, F] -- the signals this rank owns
    encode_fn       : callable(sequences, D, **kwargs) -> object with .coefficients (list of csc),
                      .residuals, .events, .stats, .energies; defaults to the GPU engine
                      (ConvolutionalMatchingPursuit.computeCoefficientsBatch on cuda:LOCAL_RANK)
    Returns a dict; with gather=True every rank gets the results of ALL signals in rank order:
      'events'   list of (t, k, c) per signal (selection order)
      'stats'    int32 [B_total, 8]
      'energies' float64 [B_total, 2]
      'coefficients' list of csc_matrix     'residuals' [B_total, T(,F)]
    """
    dist = _dist()
    if encode_fn is None:
        import os
        from .modeling import ConvolutionalMatchingPursuit
        cmp = ConvolutionalMatchingPursuit(device=int(os.environ.get('LOCAL_RANK', '0')))
        encode_fn = cmp.computeCoefficientsBatch
    res = encode_fn(sequences_shard, D, **kwargs)
    local = dict(events=res.events, stats=np.asarray(res.stats), energies=np.asarray(res.energies),
                 coefficients=res.coefficients, residuals=np.asarray(res.residuals))
    if not gather or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)          # per-signal results only: KBs per signal
    out = dict(events=[], coefficients=[])
    for part in parts:
        out['events'].extend(part['events'])
        out['coefficients'].extend(part['coefficients'])
    out['stats'] = np.concatenate([p['stats'] for p in parts], axis=0)
    out['energies'] = np.concatenate([p['energies'] for p in parts], axis=0)
    out['residuals'] = np.concatenate([p['residuals'] for p in parts], axis=0)
    return out
