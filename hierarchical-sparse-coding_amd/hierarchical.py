"""Hierarchical (multilevel) convolutional sparse coding on top of the GPU matching pursuit.

Reference surface (hsc/modeling.py): HierarchicalConvolutionalMatchingPursuit (:1427-1654) --
per-level loop, distributed-coefficient post-processing, residual through the input-level
representations -- and HierarchicalConvolutionalSparseCoder (:1671-1705).

Every level runs the same engine as the single-level coder (hsc_amd.modeling): level l >= 1
encodes the previous level's coefficient matrix [T, K_{l-1}] (float64, features = atoms of the level
below) with the level's raw dictionary [K_l, W_l, K_{l-1}], singleton atoms down-weighted by
`singletonWeight` in the selection score (:1448-1450).
"""
import collections.abc
import copy
import os

import numpy as np
import scipy.sparse

from .modeling import ConvolutionalMatchingPursuit, ConvolutionalSparseCoder, SparseApproximator, reconstructSignal


def _is_multilevel_dict(obj):
    return all(hasattr(obj, a) for a in ('getNbLevels', 'getRawDictionary', 'getBaseDictionary',
                                         'getMultiscaleDictionaries', 'countsNoSingletons'))


class HierarchicalConvolutionalMatchingPursuit(SparseApproximator):
    """hsc/modeling.py:1427-1654.  `method`: 'cmp' (persistent-kernel greedy engine) or 'locomp' (the
    reference's default: host loop over the GPU entry points, hsc_amd.locomp); the MPTK bindings
    ('mptk-mp', 'mptk-cmp') are not provided."""

    def __init__(self, method='locomp', device=0):
        self.method = method
        self.device = device

    def _level_coder(self, D):
        if self.method == 'cmp':
            return ConvolutionalSparseCoder(D, ConvolutionalMatchingPursuit(device=self.device))
        if self.method == 'locomp':
            from .locomp import LoCOMP
            # (the reference's own pseudo-inverse per group: see LoCOMP.__init__; computeCoefficientsBatch runs the device loop)
            return ConvolutionalSparseCoder(D, LoCOMP(device=self.device, refit='host'))
        if self.method in ('mptk-mp', 'mptk-cmp'):
            raise NotImplementedError("method='%s' needs the external MPTK toolkit, which this engine does not bind; "
                                      "use method='cmp' or 'locomp'" % self.method)
        raise Exception('Unsupported sparse coding method: %s' % (self.method))

    def _encode_levels(self, input, coefficients, fromLevel, multilevelDict, toleranceSnr, nbBlocks, singletonWeight):
        """Levels fromLevel .. last (:1432-1492 / :1494-1554)."""
        for level in range(fromLevel, multilevelDict.getNbLevels()):
            if toleranceSnr is not None and isinstance(toleranceSnr, collections.abc.Iterable):
                targetSnr = toleranceSnr[level]
            else:
                targetSnr = toleranceSnr
            D = multilevelDict.getRawDictionary(level)
            # the first D.shape[0] - countsNoSingletons[level] atoms are singletons (:1448-1450)
            nbSingletons = D.shape[0] - multilevelDict.countsNoSingletons[level]
            weights = np.ones((D.shape[0],), dtype=D.dtype)
            weights[:nbSingletons] = singletonWeight
            levelCoder = self._level_coder(D)
            levelCoefficients, _ = levelCoder.encode(np.asarray(input), toleranceSnr=targetSnr, nbBlocks=nbBlocks, weights=weights)
            input = levelCoefficients.toarray()                     # :1489 (todense), next level's [T, K_l] input
            coefficients.append(levelCoefficients)
        return coefficients

    def _forwardPhase(self, sequence, multilevelDict, toleranceSnr=None, nbBlocks=1, singletonWeight=0.5, stopCondition=None):
        return self._encode_levels(sequence, [], 0, multilevelDict, toleranceSnr, nbBlocks, singletonWeight)

    def _forwardPhaseFromLevel(self, sequence, coefficients, multilevelDict, toleranceSnr=None, nbBlocks=1,
                               singletonWeight=0.5, stopCondition=None):
        return self._encode_levels(coefficients[-1].toarray(), coefficients, len(coefficients), multilevelDict,
                                   toleranceSnr, nbBlocks, singletonWeight)

    def convertToDistributedCoefficients(self, coefficients):
        """hsc/modeling.py:1556-1594: the singleton columns of the LAST level are handed back to the
        level they pass through, everything else stays in the last level."""
        last = coefficients[-1].copy()
        if scipy.sparse.issparse(last):
            last = last.tocsc()
        out = []
        for level in range(len(coefficients)):
            if level < len(coefficients) - 1:
                nbFeatures = coefficients[level].shape[1]
                if scipy.sparse.issparse(coefficients[level]):
                    levelCoefficients = last[:, :nbFeatures]
                    last = scipy.sparse.hstack((scipy.sparse.csc_matrix((last.shape[0], nbFeatures), dtype=last.dtype),
                                                last[:, nbFeatures:]))
                    levelCoefficients.eliminate_zeros()
                else:
                    levelCoefficients = np.copy(last[:, :nbFeatures])
                    last[:, :nbFeatures] = 0.0
            else:
                levelCoefficients = last
            out.append(levelCoefficients)
        assert len(out) == len(coefficients)
        if scipy.sparse.issparse(coefficients[-1]):
            assert np.sum([c.nnz for c in out]) == coefficients[-1].nnz
        return out

    def _calculateResidual(self, sequence, coefficients, multilevelDict):
        """hsc/modeling.py:1596-1611"""
        baseDict = multilevelDict.getBaseDictionary()
        if baseDict.ndim == 2:
            reconstruction = np.zeros((coefficients[0].shape[0],), dtype=coefficients[0].dtype)
        else:
            reconstruction = np.zeros((coefficients[0].shape[0], baseDict.shape[-1]), dtype=coefficients[0].dtype)
        representations = multilevelDict.getMultiscaleDictionaries()
        for level in range(multilevelDict.getNbLevels()):
            reconstruction += reconstructSignal(coefficients[level], representations[level])
        return sequence - reconstruction

    def _postprocessCoefficients(self, coefficients, multilevelDict, returnDistributed=True):
        """hsc/modeling.py:1613-1634"""
        if returnDistributed:
            return self.convertToDistributedCoefficients(coefficients)
        out = []
        for level in range(multilevelDict.getNbLevels()):
            c = coefficients[level]
            if level < multilevelDict.getNbLevels() - 1:      # keep the last level only
                c = scipy.sparse.csc_matrix(c.shape, dtype=c.dtype) if scipy.sparse.issparse(c) else np.zeros_like(c)
            out.append(c)
        return out

    def computeCoefficients(self, sequence, multilevelDict, nbNonzeroCoefs=None, toleranceResidualScale=None, toleranceSnr=None,
                            nbBlocks=1, minCoefficients=None, singletonWeight=0.5, returnDistributed=True, stopCondition=None):
        """hsc/modeling.py:1636-1643"""
        assert _is_multilevel_dict(multilevelDict)
        coefficients = self._forwardPhase(sequence, multilevelDict, toleranceSnr, nbBlocks, singletonWeight, stopCondition)
        coefficients = self._postprocessCoefficients(coefficients, multilevelDict, returnDistributed)
        residual = self._calculateResidual(sequence, coefficients, multilevelDict)
        return coefficients, residual

    def _level_engines(self, nbLevels):
        from . import _native
        engines = self.__dict__.setdefault('_engines', [])
        while len(engines) < nbLevels:
            engines.append(_native.Engine(self.device))
        return engines[:nbLevels]

    def close(self):
        """Release the per-level GPU engines (and their workspaces) of computeCoefficientsBatch."""
        for e in self.__dict__.pop('_engines', []):
            e.close()

    def computeCoefficientsBatch(self, sequences, multilevelDict, toleranceSnr=None, nbBlocks=1, singletonWeight=0.5,
                                 returnDistributed=True, chained=True, memoryBudget=None, epilogue='device', returnEvents=False,
                                 deviceInput=None, residuals='samples'):
        """Batch form (the reference has no batch axis): `sequences` [B,T] (or [B,T,F]); every level encodes
        many signals per GPU call.  chained=True keeps the level hand-off on the device
        (hscmp_encode_batch_from_level): the dense [T, K_prev] float64 input of a level (modeling.py:1489)
        is scattered from the previous level's coefficient slots in GPU memory, in chunks of signals
        that fit `memoryBudget` bytes (default: 60 % of the GPU's memory) -- at BASELINE config 4 that input is 134 MB per signal and can
        not travel through the host.  chained=False builds the dense inputs on the host (small cases).
        epilogue='device' (chained only): redistribution of the singleton columns (:1556-1594), CSC assembly (:1171-1181),
        residual through the input-level representations (:1596-1611) and the event records (dataset.py:798-811) come
        from ONE kernel per chunk over the last level's device-resident coefficient slots (hscmp_hierarchy_epilogue) and
        are fetched once; 'host' assembles them signal by signal on the CPU (same results bit for bit).
        deviceInput: device address of `sequences` ([B,T,F] in the level-0 compute dtype) when they already sit in GPU
        memory (chained only); the host array is then used for its shape and dtype only -- whatever falls back to the host
        (a signal the device loop gave up on, an epilogue outside the kernel's key layout) copies the signals it needs back
        from the device (hscmp_copy_from_device).
        residuals='energy' (device epilogue only): the second item returned is the float64 vector [B] of residual energies
        (sum of squares, summed on the device); the residual samples -- T float64 per signal -- never cross PCIe.
        Returns (per-signal lists of per-level coefficient matrices, residuals [B,T(,F)] float64,
        per-level kernel timings); with returnEvents=True a fourth item: per-signal event record arrays.
        The steps live in _LevelPipeline (below): level set-up, one level over one chunk with event-list regrowth, chunk sizing,
        the two epilogues, the per-signal fallback."""
        assert residuals in ('samples', 'energy')
        assert _is_multilevel_dict(multilevelDict)
        if self.method not in ('cmp', 'locomp'):
            raise Exception('Unsupported sparse coding method: %s' % (self.method))
        pipe = _LevelPipeline(self, sequences, multilevelDict, toleranceSnr, nbBlocks, singletonWeight, returnDistributed,
                              epilogue if chained else 'host', returnEvents, deviceInput if chained else None, residuals, memoryBudget)
        return pipe.run_chained() if chained else pipe.run_unchained()

    def _host_epilogue_chunk(self, engines, first, count, nbLevels, multilevelDict, returnDistributed, sequences, results, returnEvents,
                             residual_out, energy_out):
        """The chunk's epilogue on the host (:1556-1634, :1596-1611) from the slot lists of every level's engine -- level 0 holds
        the whole batch, the levels above it this chunk.  Same results as the device epilogue (tests/test_hierarchical.py)."""
        from . import _native
        from .modeling import _slots_to_csc
        levels = []
        for l in range(nbLevels):
            eng, off = engines[l], (first if l == 0 else 0)
            st, sk, sa = eng.fetch_slots()
            stats = eng.fetch_stats()
            T = eng._batch[1]
            levels.append([_slots_to_csc(st[off + b], sk[off + b], sa[off + b], int(stats[off + b, _native.STAT_SLOTS]), (T, eng.K), 1e-16)
                           for b in range(count)])
        for b in range(count):
            cb = self._postprocessCoefficients([levels[l][b] for l in range(nbLevels)], multilevelDict, returnDistributed)
            residual = np.asarray(self._calculateResidual(sequences[first + b], cb, multilevelDict), dtype=np.float64)
            if residual_out is not None:
                residual_out[b] = residual.reshape(residual_out[b].shape)
            if energy_out is not None:
                energy_out[b] = float(np.sum(np.square(residual)))
            ev = None
            if returnEvents:
                from .dataset import convertSparseMatricesToEvents
                ev = convertSparseMatricesToEvents(cb)
            results[first + b] = (cb, None, ev)

    def _device_epilogue(self, engines, first, count, nbLevels, multilevelDict, returnDistributed, slot_counts, results, returnEvents, residual_out,
                         energy_out=None):
        """hscmp_hierarchy_epilogue for one chunk: per signal the per-level coefficient matrices (:1556-1634), the residual
        (:1596-1611) and the event records (dataset.py:798-811), from the last level's device-resident slots."""
        import scipy.sparse
        last = engines[nbLevels - 1]
        reps = multilevelDict.getMultiscaleDictionaries()
        counts = [int(multilevelDict.getRawDictionary(l).shape[0]) for l in range(nbLevels)]
        levels = []
        for l in range(nbLevels):
            if returnDistributed:
                c0 = counts[l - 1] if l > 0 else 0         # level l owns the columns its own (non-singleton) atoms sit in
                levels.append((c0, counts[l], reps[l]))
            else:
                levels.append((0, counts[l], reps[l]) if l == nbLevels - 1 else (0, 0, None))
        n, colptr, offsets, indices, data, events, residual = last.hierarchy_epilogue(
            engines[0], first if nbLevels > 1 else 0, levels, 1e-16, slot_counts, want_events=returnEvents, want_residual=residual_out is not None,
            residual_out=residual_out, energy_out=energy_out)
        T = last._batch[1]
        # per-level column pointers of the whole chunk at once: level l is the slice [c0, c1) of the last level's columns
        ptrs, starts = [], []
        for l, (c0, c1, _) in enumerate(levels):
            if c1 <= c0:
                ptrs.append(None); starts.append(None)
                continue
            base = colptr[:, c0:c0 + 1]
            p = np.clip(colptr[:, :counts[l] + 1] - base, 0, None)
            np.minimum(p, (colptr[:, c1:c1 + 1] - base), out=p)
            ptrs.append(np.ascontiguousarray(p, dtype=np.int32)); starts.append(base[:, 0])
        for b in range(count):
            o = int(offsets[b])
            mats = []
            for l in range(nbLevels):
                if ptrs[l] is None:
                    mats.append(scipy.sparse.csc_matrix((T, counts[l]), dtype=np.float64))
                    continue
                lo = o + int(starts[l][b]); hi = lo + int(ptrs[l][b, -1])
                mats.append(_csc_from_checked_arrays(data[lo:hi], indices[lo:hi], ptrs[l][b], (T, counts[l])))
            ev = events[o:o + int(n[b])] if events is not None else None
            results[first + b] = (mats, None, ev)

    def computeCoefficientsFromLevel(self, sequence, coefficients, multilevelDict, nbNonzeroCoefs=None, toleranceResidualScale=None,
                                     toleranceSnr=None, nbBlocks=1, minCoefficients=None, singletonWeight=0.5, stopCondition=None,
                                     returnDistributed=True):
        """hsc/modeling.py:1645-1654"""
        assert _is_multilevel_dict(multilevelDict)
        coefficients = copy.deepcopy(coefficients)
        coefficients = self._forwardPhaseFromLevel(sequence, coefficients, multilevelDict, toleranceSnr, nbBlocks,
                                                   singletonWeight, stopCondition)
        return self._postprocessCoefficients(coefficients, multilevelDict, returnDistributed)


class _LevelPipeline(object):
    """One call of HierarchicalConvolutionalMatchingPursuit.computeCoefficientsBatch, step by step (hsc/modeling.py:1432-1492 per
    level, :1556-1634 / :1596-1611 afterwards).  method='locomp' (the reference's default, :1429) is the same pipeline with every
    level's engine in the LoCOMP loop (csrc/hscmp_locomp.h).

      level_setup      dictionary, weights (:1448-1450), SNR target (:1439-1442) of a level
      encode_level     one level over a batch / chunk, with the event-capacity regrowth (hscmp_grow_events + hscmp_continue)
      chunk_size       signals per chunk of the levels >= 1 (their dense float64 input [T, F_l] lives on the device)
      run_chunk_levels levels 1 .. last over one chunk (device-chained hand-off)
      device_epilogue / host_epilogue   redistribution, CSC, events, residual of a chunk
      fallback_signal  a signal the device loop gave up on (stop reason 'group'): the per-signal entry
      host_signal      signal b on the host, from the device when the batch was handed over as a device pointer"""

    def __init__(self, hcmp, sequences, multilevelDict, toleranceSnr, nbBlocks, singletonWeight, returnDistributed, epilogue, returnEvents,
                 deviceInput, residuals, memoryBudget):
        self.hcmp, self.sequences, self.mld = hcmp, sequences, multilevelDict
        self.toleranceSnr, self.nbBlocks, self.singletonWeight = toleranceSnr, nbBlocks, singletonWeight
        self.returnDistributed, self.returnEvents, self.deviceInput, self.residuals = returnDistributed, returnEvents, deviceInput, residuals
        self.memoryBudget = memoryBudget
        self.device_epilogue_on = epilogue == 'device'
        assert residuals == 'samples' or self.device_epilogue_on, "residuals='energy' needs the device epilogue"
        self.locomp = hcmp.method == 'locomp'
        self.nbLevels = multilevelDict.getNbLevels()
        self.B, self.T = sequences.shape[0], sequences.shape[1]
        self.group_failed = set()
        self.per_level = [[None] * self.B for _ in range(self.nbLevels)]
        self.timings = []
        self.engines = None
        self.dt0 = None

    # ---- per level
    def level_setup(self, level):
        """(D, weights, targetSnr, eps) of a level: :1439-1450"""
        snr = self.toleranceSnr
        targetSnr = snr[level] if (snr is not None and isinstance(snr, collections.abc.Iterable)) else snr
        D = self.mld.getRawDictionary(level)
        nbSingletons = D.shape[0] - self.mld.countsNoSingletons[level]
        weights = np.ones((D.shape[0],), dtype=D.dtype)
        weights[:nbSingletons] = self.singletonWeight
        return D, weights, targetSnr, float(np.finfo(D.dtype).eps)

    def start_capacity(self, eng, maxEvents):
        """An engine remembers the event capacity its last batch ended with: a stream of similar batches -- the bench, a dataset --
        then starts with lists that fit, instead of growing them twice per level and batch."""
        from . import _native
        hint = getattr(eng, '_capacity_hint', None)
        if hint is not None and hint[0] == (self.T, self.nbBlocks):
            maxEvents = min(max(maxEvents, hint[1]), _native.max_event_capacity(self.T))
        return maxEvents

    def regrow_until_done(self, eng, maxEvents, kernel_ms):
        """Signals that stopped for want of event capacity: enlarge the lists in place and resume (exact: a round is never started
        unless all its atoms fit).  Returns (stats, final capacity)."""
        from . import _native
        while True:
            stats = eng.fetch_stats()
            if not np.any(stats[:, _native.STAT_STOP] == _native.STOP_CAPACITY):
                return stats, maxEvents
            bound = _native.max_event_capacity(self.T)
            if maxEvents >= bound:
                raise _native.HscmpError('the pursuit does not converge: more than %d selections per signal without meeting a stop '
                                         'rule (the same atoms are re-selected; the reference would not terminate)' % maxEvents)
            maxEvents = min(4 * maxEvents, bound)
            eng.grow_events(maxEvents)
            eng.continue_rounds(0)
            kernel_ms[2] += float(eng.last_kernel_ms()[2])

    def encode_level(self, eng, encode, count, targetSnr, eps, maxEvents=4096, lazy=False, offset=0):
        """encode(params) on `eng`, with regrowth; returns (per-signal coefficients or None when they stay on the device,
        timing dict, stats)"""
        from . import _native
        from .modeling import _slots_to_csc
        maxEvents = self.start_capacity(eng, maxEvents)
        params = _native.make_params(None, None, targetSnr, self.nbBlocks, 1e-16, eps, maxEvents, 0)
        encode(params)
        kernel_ms = [float(v) for v in eng.last_kernel_ms()]
        stats, maxEvents = self.regrow_until_done(eng, maxEvents, kernel_ms)
        eng._capacity_hint = ((self.T, self.nbBlocks), maxEvents)
        self.group_failed.update(int(b) + offset for b in np.where(stats[:, _native.STAT_STOP] == _native.STOP_GROUP)[0])
        K = eng.K
        if self.device_epilogue_on:
            out = None                                     # the coefficient slots stay on the device (hscmp_hierarchy_epilogue)
        else:
            st, sk, sa = eng.fetch_slots()
            if lazy:
                # (the CSC assembly joins the per-signal host epilogue, which runs on all cores)
                out = [(st[b], sk[b], sa[b], int(stats[b, _native.STAT_SLOTS]), (self.T, K), 1e-16) for b in range(count)]
            else:
                out = [_slots_to_csc(st[b], sk[b], sa[b], int(stats[b, _native.STAT_SLOTS]), (self.T, K), 1e-16) for b in range(count)]
        tm = dict(variant=eng.last_variant(), kernel_ms=kernel_ms,
                  selections=int(stats[:, _native.STAT_ITERATIONS].sum()), duplicates=int(stats[:, _native.STAT_DUPLICATES].sum()),
                  rounds=int(stats[:, _native.STAT_ROUNDS].sum()),
                  stops={_native.STOP_NAMES.get(int(c), str(c)): int(n_) for c, n_ in zip(*np.unique(stats[:, _native.STAT_STOP], return_counts=True))})
        return out, tm, stats

    # ---- host copies of the inputs
    def host_signal(self, b):
        """Signal b as a host array: from the device when the batch came as a device pointer (the host array may be a placeholder)."""
        seq = np.asarray(self.sequences)
        if self.deviceInput is None:
            return seq[b]
        shape = seq.shape[1:]
        nbytes = int(np.prod(shape)) * np.dtype(self.dt0).itemsize
        return self.engines[0].copy_from_device(int(self.deviceInput) + b * nbytes, shape, self.dt0)

    def fallback_signal(self, b):
        """(coefficients, residual) of signal b through the per-signal entry (host-loop methods have no group limit)."""
        return self.hcmp.computeCoefficients(self.host_signal(b), self.mld, toleranceSnr=self.toleranceSnr, nbBlocks=self.nbBlocks,
                                             singletonWeight=self.singletonWeight, returnDistributed=self.returnDistributed)

    # ---- chained = False: dense level inputs built on the host (small cases)
    def run_unchained(self):
        inputs = np.asarray(self.sequences)
        for level in range(self.nbLevels):
            D, weights, targetSnr, _ = self.level_setup(level)
            if self.locomp:
                from .locomp import LoCOMP
                cmp = LoCOMP(device=self.hcmp.device)
            else:
                cmp = ConvolutionalMatchingPursuit(device=self.hcmp.device)
            res = cmp.computeCoefficientsBatch(inputs, D, toleranceSnr=targetSnr, nbBlocks=self.nbBlocks, weights=weights)
            self.per_level[level] = res.coefficients
            # (a batch that ran on the host loop -- HSCMP_LOCOMP_HOST -- keeps neither counters nor kernel times)
            self.timings.append(dict(level=level, variant=res.variant,
                                     kernel_ms=[float(v) for v in res.kernel_ms] if res.kernel_ms is not None else [0.0, 0.0, 0.0, 0.0],
                                     selections=int(res.stats[:, 4].sum()) if res.stats is not None else 0))
            if level + 1 < self.nbLevels:
                inputs = np.stack([c.toarray() for c in res.coefficients], axis=0)      # [B, T, K_level] float64
        return self.finish_on_host()

    # ---- chained = True
    def chunk_size(self, setups, stats0):
        """Per signal on the device, for EVERY level >= 1 at once (each level's engine keeps its workspace while the chunk moves up
        the hierarchy): the dense float64 residual [T, F_l] (the input is scattered straight into it), per-row state, and the event
        / slot / hash lists (about 80 bytes per list entry, sized from the level-0 counts)."""
        from . import _native
        if self.nbLevels == 1:
            return self.B
        budget = self.memoryBudget
        if budget is None:
            budget = 0.6 * self.engines[0].mem_info()[1]      # (of the total: the cached engines already hold their workspaces)
        nin0 = int(stats0[:, _native.STAT_SLOTS].max()) if self.B else 0
        per_signal = sum(1.05 * self.T * setups[l][0].shape[2] * 8 + 160 * self.T + 80.0 * max(4096, 2 * nin0) for l in range(1, self.nbLevels))
        return int(max(1, min(self.B, budget // max(per_signal, 1.0))))

    def run_chunk_levels(self, setups, first, count, stats0):
        """Levels 1 .. last over signals [first, first + count); returns the last level's stats."""
        from . import _native
        engines = self.engines
        last_stats = stats0
        for l in range(1, self.nbLevels):
            _, _, targetSnr, eps = setups[l]
            prev, pfirst = (engines[0], first) if l == 1 else (engines[l - 1], 0)
            # every input non-zero is explained at least once (by its singleton): size the lists for that
            nin = int(last_stats[pfirst:pfirst + count, _native.STAT_SLOTS].max())
            coefs, tm, last_stats = self.encode_level(
                engines[l], lambda p, e=engines[l], pv=prev, pf=pfirst: e.encode_batch_from_level(pv, pf, count, 1e-16, p),
                count, targetSnr, eps, maxEvents=max(4096, nin + nin // 4 + 64), lazy=True, offset=first)
            if coefs is not None:
                self.per_level[l][first:first + count] = coefs
            acc = self.timings[l]
            for name_, n_ in tm['stops'].items():
                acc.setdefault('stops', {})[name_] = acc.setdefault('stops', {}).get(name_, 0) + n_
            acc['variant'] = tm['variant']; acc['selections'] += tm['selections']; acc['duplicates'] += tm['duplicates']; acc['rounds'] += tm['rounds']; acc['chunks'] += 1
            acc['kernel_ms'] = [a + b for a, b in zip(acc['kernel_ms'], tm['kernel_ms'])]
        return last_stats

    def epilogue_chunk(self, first, count, stats0, last_stats, results, residual_all, energy_all):
        from . import _native
        nbLevels = self.nbLevels
        o = first if nbLevels == 1 else 0
        slot_counts = (stats0 if nbLevels == 1 else last_stats)[o:o + count, _native.STAT_SLOTS]
        res_out = None if residual_all is None else residual_all[first:first + count]
        en_out = None if energy_all is None else energy_all[first:first + count]
        try:
            self.hcmp._device_epilogue(self.engines, first, count, nbLevels, self.mld, self.returnDistributed, slot_counts, results,
                                       self.returnEvents, res_out, en_out)
        except _native.HscmpError as ex:
            # a shape outside the epilogue kernel's key layout (2^20 atoms / list entries, 2^24 samples): this chunk's
            # redistribution and residual are done on the host from the fetched slot lists, as with epilogue='host'
            if ex.code != _native.ERR_UNSUPPORTED:
                raise
            seqs = _ChunkSignals(self, first)
            self.hcmp._host_epilogue_chunk(self.engines, first, count, nbLevels, self.mld, self.returnDistributed, seqs, results,
                                           self.returnEvents, res_out, en_out)

    def run_chained(self):
        from . import _native
        from .modeling import _compute_dtype
        B, T, nbLevels = self.B, self.T, self.nbLevels
        sequences = self.sequences
        engines = self.engines = self.hcmp._level_engines(nbLevels)      # kept across calls: their workspaces are tens of GB
        for e in engines:
            e.set_method(_native.METHOD_LOCOMP if self.locomp else _native.METHOD_CMP)
        # level 0: the signals themselves, all B at once
        D, weights, targetSnr, eps = self.level_setup(0)
        dt = self.dt0 = _compute_dtype(sequences.dtype, D.dtype)
        D3 = np.ascontiguousarray(D.reshape((D.shape[0], D.shape[1], -1)), dtype=dt)
        engines[0].set_dictionary(D3, np.asarray(weights, dtype=dt))
        nfeat = int(np.prod(np.asarray(sequences).shape[2:])) if np.asarray(sequences).ndim > 2 else 1
        if self.deviceInput is not None:
            assert np.asarray(sequences).dtype == dt, 'deviceInput must hold the level-0 compute dtype'
            enc0 = lambda p: engines[0].encode_batch_device(int(self.deviceInput), B, T, p)
        else:
            x = np.ascontiguousarray(np.asarray(sequences).reshape((B, T, -1)), dtype=dt)
            enc0 = lambda p: engines[0].encode_batch(x, p)
        self.per_level[0], tm, stats0 = self.encode_level(engines[0], enc0, B, targetSnr, eps, lazy=True)
        tm['level'] = 0
        self.timings.append(tm)
        setups = [None] + [self.level_setup(l) for l in range(1, nbLevels)]
        for l in range(1, nbLevels):
            engines[l].set_dictionary(setups[l][0], setups[l][1], dtype=np.float64)
            self.timings.append(dict(level=l, variant='', kernel_ms=[0.0, 0.0, 0.0, 0.0], selections=0, duplicates=0, rounds=0, chunks=0))
        chunk = self.chunk_size(setups, stats0)
        dev_epi = self.device_epilogue_on
        results = [None] * B
        residual_all = np.empty((B, T, nfeat), dtype=np.float64) if (dev_epi and self.residuals == 'samples') else None
        energy_all = np.empty((B,), dtype=np.float64) if (dev_epi and self.residuals == 'energy') else None
        first = 0
        while first < B and (nbLevels > 1 or dev_epi):
            count = min(chunk, B - first)
            try:
                last_stats = self.run_chunk_levels(setups, first, count, stats0)
            except _native.HscmpError as ex:
                # out of device memory part-way through a chunk (the budget is an estimate): halve the chunk, run it again
                if ex.code == _native.ERR_ALLOC and count > 1:
                    chunk = max(1, count // 2)      # (the timing totals keep what the failed attempt had already run)
                    continue
                raise
            if dev_epi:
                self.epilogue_chunk(first, count, stats0, last_stats, results, residual_all, energy_all)
            first += count
        if not dev_epi:
            return self.finish_on_host()
        for b in sorted(self.group_failed):
            cb, res_b = self.fallback_signal(b)
            ev = None
            if self.returnEvents:
                from .dataset import convertSparseMatricesToEvents
                ev = convertSparseMatricesToEvents(cb)
            results[b] = (cb, None, ev)
            if residual_all is not None:
                residual_all[b] = np.asarray(res_b, dtype=np.float64).reshape(residual_all[b].shape)
            if energy_all is not None:
                energy_all[b] = float(np.sum(np.square(np.asarray(res_b, dtype=np.float64))))
        if energy_all is not None:
            second = energy_all
        else:
            second = residual_all[:, :, 0] if np.asarray(sequences).ndim == 2 else residual_all
        out = ([r[0] for r in results], second, self.timings)
        return out + ([r[2] for r in results],) if self.returnEvents else out

    def finish_on_host(self):
        """Host epilogue per signal (redistribution :1556-1594, residual :1596-1611), spread over the cores: the numpy kernels it
        spends its time in release the interpreter lock."""
        from .modeling import _slots_to_csc
        nbLevels, B = self.nbLevels, self.B

        def finish(b):
            if b in self.group_failed:
                return self.fallback_signal(b)
            levels = [self.per_level[l][b] for l in range(nbLevels)]
            levels = [_slots_to_csc(*c) if isinstance(c, tuple) else c for c in levels]
            cb = self.hcmp._postprocessCoefficients(levels, self.mld, self.returnDistributed)
            return cb, self.hcmp._calculateResidual(self.host_signal(b), cb, self.mld)
        workers = max(1, min(int(os.environ.get('HSC_EPILOGUE_WORKERS', '16')), os.cpu_count() or 1, B))
        if workers > 1 and self.deviceInput is None:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=workers) as pool:
                done = list(pool.map(finish, range(B)))
        else:
            done = [finish(b) for b in range(B)]         # (device copies go through one engine: one thread)
        coefficients = [d[0] for d in done]
        out = (coefficients, np.stack([d[1] for d in done], axis=0), self.timings)
        if self.returnEvents:
            from .dataset import convertSparseMatricesToEvents
            out = out + ([convertSparseMatricesToEvents(c) for c in coefficients],)
        return out


class _ChunkSignals(object):
    """sequences[i] for the host epilogue of a chunk: host rows, or copies from the device when the batch came as a device pointer."""

    def __init__(self, pipe, first):
        self.pipe = pipe

    def __getitem__(self, i):
        return self.pipe.host_signal(int(i))


def _csc_from_checked_arrays(data, indices, indptr, shape):
    """csc_matrix over arrays the device epilogue wrote (canonical: int32 indices sorted inside every column, no
    duplicates, no zeros).  The public constructor re-validates and re-derives the index dtype -- 18 us per matrix,
    36 ms for the 2048 matrices of a config-4 batch; a matrix whose attributes are set directly is the same object
    to every consumer.  Falls back to the constructor if this scipy lays the object out differently."""
    import scipy.sparse
    try:
        m = scipy.sparse.csc_matrix.__new__(scipy.sparse.csc_matrix)
        m.data, m.indices, m.indptr = data, indices, indptr
        m._shape = (int(shape[0]), int(shape[1]))
        m.maxprint = 50
        m.has_sorted_indices = True
        m.has_canonical_format = True
        if m.shape != m._shape or m.nnz != len(data):
            raise AttributeError
        return m
    except Exception:
        return scipy.sparse.csc_matrix((data, indices, indptr), shape=shape, copy=False)


class HierarchicalConvolutionalSparseCoder(object):
    """hsc/modeling.py:1671-1705"""

    def __init__(self, multilevelDict, approximator):
        assert _is_multilevel_dict(multilevelDict)
        if not multilevelDict.hasSingletonBases:
            multilevelDict = multilevelDict.withSingletonBases()
        self.multilevelDict = multilevelDict
        self.approximator = approximator

    def encode(self, sequence, *args, **kwargs):
        assert sequence.ndim == 1 or sequence.ndim == 2
        return self.approximator.computeCoefficients(sequence, self.multilevelDict, *args, **kwargs)

    def encodeFromLevel(self, sequence, coefficients, *args, **kwargs):
        assert len(coefficients) > 0
        return self.approximator.computeCoefficientsFromLevel(sequence, coefficients, self.multilevelDict, *args, **kwargs)

    def reconstruct(self, coefficients):
        assert len(coefficients) > 0
        baseDict = self.multilevelDict.getBaseDictionary()
        if baseDict.ndim == 2:
            signal = np.zeros((coefficients[0].shape[0],), dtype=coefficients[0].dtype)
        else:
            signal = np.zeros((coefficients[0].shape[0], baseDict.shape[-1]), dtype=coefficients[0].dtype)
        representations = self.multilevelDict.getMultiscaleDictionaries()
        for level in range(self.multilevelDict.getNbLevels()):
            signal += reconstructSignal(coefficients[level], representations[level])
        return signal
