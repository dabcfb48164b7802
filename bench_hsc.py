"""bench.py --config 4 | 5: the hierarchical encoder (hsc/modeling.py:1427-1654; the per-level loop of
scripts/learn_mlcsc_dataset.py:108-133) on BASELINE.json's multilevel configurations.

  config 4  2 levels at the dimensions of configs[3]: level 0 256 atoms x 64 taps, level 1 (256 singletons + 128) atoms
            x 17 taps x 256 features on the level-0 coefficient streams; 1024 signals of 65536 samples per GPU;
            toleranceSnr [30, 40] dB, nbBlocks=10, singletonWeight 0.95 (learn_mlcsc_dataset.py:113).  The hierarchy is
            hsc_amd.synth.make_hierarchy: level-1 atoms are compositions of level-0 atoms, signals are rendered from both
            levels, so the encode reconstructs its input (checked: SNR >= 25 dB on every signal).  17 taps, not 16: with
            scales [64, 79] the reference's own centre conventions reconstruct every level-1 pattern one sample late
            (see make_hierarchy); --level1-taps 16 runs the literal shape (same cost, no SNR check).
  config 5  3 levels in the style of configs[4]: generated Perlin dictionary, scales [32, 64, 128] (taps [32, 33, 65]),
            4x overcomplete per level plus singleton bases, Poisson-event signals at compression ratio 0.25
            (scripts/generate_dataset.py:57-94); 128 signals of 65536 samples per GPU; toleranceSnr [30, 35, 35].

One step = one multilevel encode of the rank's batch, signals resident in HBM when it starts: level 0 (f32 MFMA kernels,
blocked selection), every further level on the previous level's coefficient slots scattered on the device (float64 sparse
x sparse kernels), then the device epilogue (redistribution, CSC, residual) and the fetch of the per-signal results.
`value` = atom selections of all levels per second of that whole step.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK_FP32_MFMA_TFLOPS = 157.3
SNR4, SNR5 = [30.0, 40.0], [30.0, 35.0, 35.0]


def build_workload(config, B, T, first, level1_taps=17, only_dictionary=False):
    """(multilevel dictionary with singleton bases, signals [B,T] float32, encode kwargs, description)"""
    import hsc_amd.synth as synth
    if config == 4:
        mld = synth.make_hierarchy(W1=level1_taps, seed=4)
        xs = None if only_dictionary else synth.make_hierarchy_batch(mld, T, first, B, seed=4)
        kw = dict(toleranceSnr=SNR4, nbBlocks=10, singletonWeight=0.95)
        desc = ('BASELINE configs[3]: 2-level HSC, L1 256x64 then L2 (256 singletons + 128)x%dx256 on the L1 coefficient streams, '
                '%d signals/GPU x len %d, toleranceSnr %s, nbBlocks=10, singletonWeight=0.95' % (level1_taps, B, T, SNR4))
    else:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import generate_dataset as gd
        mld = gd.build([32, 64, 128], 4.0, patience=100)
        xs = None
        if not only_dictionary:
            xs = gd.signals(mld, B, T, rate=5e-4, compression=0.25, seed=5, first=first)[0]
        kw = dict(toleranceSnr=SNR5, nbBlocks=10, singletonWeight=0.95)
        desc = ('BASELINE configs[4]: 3-level HSC on generated data (Perlin dictionary, scales [32,64,128], 4x overcomplete + singleton '
                'bases, Poisson events, compression 0.25), %d signals/GPU x len %d, toleranceSnr %s, nbBlocks=10' % (B, T, SNR5))
    return mld.withSingletonBases(), xs, kw, desc


# ---- roofline of the level >= 1 kernels (sparse input x sparse dictionary, float64) ------------------------------------
# What bounds them is not bandwidth but the chain of DEPENDENT steps of a round of the blocked selection: memory round trips whose
# addresses come out of the previous one (the signal's state -- its dense [T, F] float64 residual, 17-134 MB -- lives beyond the L2),
# and the list work one wave does in LDS between them.  The bound of the round-parallel loop is built from what the in-kernel stamps
# measured for exactly those two things (profiles/r03_stamps_round_parallel.txt, config 5, workgroup 0):
#   * a dependent round trip under the loop's own load (eleven waves gathering at once): the window gather of the re-correlation is two
#     of them, 10 678 cycles per round => 5 339 cycles = 2.2 us each (the idle Infinity-Cache hit of MI355X_MICROARCH.md is 227 ns: the
#     model of round 3 priced the trips at that and was missed 17 x);
#   * the serial LDS chain of one wave per round: pairing 5 194 + register sort 5 679 + chains 2 265 + per-row best 3 852 = 16 990 cycles.
# model per round = 8 trips x 2.2 us + 7.1 us x (3W - 2) / 145 (the chain scaled to the level's window); everything else a round spends (barrier waits, the candidate phase beside the gather,
# selection and prefix in wave 0, segment scans) is slack against it: `frac` = bound / measured.  The idle-latency figure stays in the
# line as `idle_latency_model`.  The HBM side is reported beside it: algorithmic bytes per applied atom against 8 TB/s.
IC_HIT_LATENCY_S = 227e-9          # MI355X_MICROARCH.md, "global_load_dword (Infinity Cache hit latency)", idle chip
CLOCK_HZ = 2.4e9
LOADED_TRIP_S = 5339 / CLOCK_HZ    # one dependent round trip under eleven gathering waves (stamps, see above)
LDS_CHAIN_S = 16990 / CLOCK_HZ     # pairing + sort + chains + per-row best of one wave, per round (stamps, see above)
LDS_CHAIN_ROWS = 3 * 49 - 2        # ... measured on windows of 3W - 2 rows with W = 33 and 65 (config 5 levels 1 / 2, both in the stamps):
                                   # the lists that chain walks grow with the rows of a window, so it is scaled by (3W - 2) / 145
PEAK_HBM_BYTES = 8.0e12
RP_ROUND_TRIPS = 8                 # round-parallel loop, per ROUND: block arg-max ends | (k, c) + row lists | span cells |
                                   # subtraction: cell + list | list append | window lists | window cells | segment scan
SEQ_ROUND_TRIPS_PER_ATOM = 6       # iterate_kernel<SparseRecorr>, per ATOM: slot probe | lists | cells | append | rows | segments


def level_roofline(Dl, T, B, tm, nbBlocks):
    K, W, F = Dl.shape
    nz = float(np.count_nonzero(Dl)) / K
    atoms = max(tm['selections'], 1)
    loop_s = 1e-3 * max(tm['kernel_ms'][2], 1e-9)
    rp = tm['variant'].endswith('_rp')
    # algorithmic bytes per applied atom: row lists (count + 8 features = 36 B) of the W span rows (energies) and of the
    # 3W-2 window rows (re-correlation), ~one listed cell per two rows (value 8 B), the atom's non-zeros (read the cell,
    # write it: 16 B each), the 2W-1 rows of per-position best written back (coefficient 8 B + atom 4 B)
    bytes_per_atom = (W + 3 * W - 2) * 36 + (4 * W - 2) // 2 * 8 + nz * 16 + (2 * W - 1) * 12
    rounds = tm.get('rounds')
    model_s = idle_s = None
    waves_of_signals = max(1.0, np.ceil(B / 256.0))   # one workgroup per CU: up to 256 signals side by side, the rest queue behind them
    if rp and rounds:
        model_s = rounds / B * (RP_ROUND_TRIPS * LOADED_TRIP_S + LDS_CHAIN_S * (3 * W - 2) / LDS_CHAIN_ROWS) * waves_of_signals
        idle_s = rounds / B * RP_ROUND_TRIPS * IC_HIT_LATENCY_S * waves_of_signals
    elif not rp:
        # the sequential loops (one team of 256 threads per signal, co-resident workgroups overlap: two per CU, four in the packed
        # build): no stamped bound yet -- the idle-latency count only, reported as such
        idle_s = atoms / B * SEQ_ROUND_TRIPS_PER_ATOM * IC_HIT_LATENCY_S * max(1.0, B / 1024.0)
    out = dict(bound='latency', kernel='iterate_rp_kernel<RpSparse>' if rp else 'iterate_kernel<LocompSparse>' if 'locomp' in tm['variant'] else 'iterate_kernel<SparseRecorr>',
               atoms_per_s=atoms / loop_s, us_per_atom_per_signal=1e6 * loop_s * B / atoms,
               dictionary_nonzeros_per_atom=nz,
               latency_model={'loaded_round_trip_s': LOADED_TRIP_S, 'lds_chain_per_round_s': LDS_CHAIN_S * (3 * W - 2) / LDS_CHAIN_ROWS,
                              'dependent_round_trips': ('%d per round' % RP_ROUND_TRIPS) if rp else ('%d per atom' % SEQ_ROUND_TRIPS_PER_ATOM),
                              'rounds_per_signal': (rounds / B) if rounds else None, 'model_ms': None if model_s is None else 1e3 * model_s,
                              'frac': None if model_s is None else model_s / loop_s,
                              'source': 'profiles/r03_stamps_round_parallel.txt (dependent trips under load, serial LDS chain of one wave)'},
               idle_latency_model={'round_trip_s': IC_HIT_LATENCY_S, 'model_ms': None if idle_s is None else 1e3 * idle_s,
                                   'frac': None if idle_s is None else idle_s / loop_s},
               hbm={'algorithmic_bytes_per_atom': bytes_per_atom, 'achieved_gb_s': bytes_per_atom * atoms / loop_s / 1e9,
                    'frac': bytes_per_atom * atoms / loop_s / PEAK_HBM_BYTES})
    return out


class _OracleLevelCoder(object):
    """cpu_baseline only: the reference's level coder restated in C (oracle/), same encode() contract."""

    def __init__(self, D):
        self.D = D

    def encode(self, X, **kw):
        from oracle import hsc_oracle as orc
        coefficients, residual, info = orc.cmp_encode(np.asarray(X), self.D, **kw)
        _OracleLevelCoder.selections += int(info['iterations'])
        return coefficients, residual

    selections = 0


def _cpu_worker(job):
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    mlds, x, kw = job
    ref = HierarchicalConvolutionalMatchingPursuit(method='cmp')
    ref._level_coder = lambda D: _OracleLevelCoder(D)
    _OracleLevelCoder.selections = 0
    t0 = time.perf_counter()
    ref.computeCoefficients(x, mlds, **kw)
    return _OracleLevelCoder.selections, time.perf_counter() - t0


def cpu_baseline_main(payload):
    """Child process (never touches the GPU): the host logic of the hierarchical encoder on the C oracle as level coder,
    one signal per worker process."""
    import multiprocessing as mp
    cfg = json.loads(payload)
    mlds, xs, kw, _ = build_workload(cfg['config'], cfg['nproc'], cfg['T'], 0, cfg.get('level1_taps', 17))
    t0 = time.perf_counter()
    jobs = [(mlds, xs[i], kw) for i in range(cfg['nproc'])]
    if cfg['nproc'] == 1:
        out = [_cpu_worker(jobs[0])]
    else:
        with mp.get_context('fork').Pool(cfg['nproc']) as pool:
            out = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    nsel = int(sum(o[0] for o in out))
    print(json.dumps({'value': nsel / wall, 'unit': 'atom-selections/s', 'cores': cfg['nproc'], 'kind': 'port',
                      'host_cores': os.cpu_count(),
                      'sample': '%d signals of the bench workload (one per process), host logic of hsc/modeling.py:1427-1654 on the C '
                                'oracle (oracle/hsc_oracle.c: materialised table, full scans) as level coder, wall %.1f s' % (cfg['nproc'], wall),
                      'per_core': nsel / wall / cfg['nproc']}))


def run(args, ctx):
    import hsc_amd.synth as synth  # noqa: F401
    from hsc_amd import _native
    from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit
    torch, dist, dev, rank, world = ctx['torch'], ctx['dist'], ctx['dev'], ctx['rank'], ctx['world']
    config = args.config
    B = args.batch or (1024 if config == 4 else 128)
    T = args.T
    steps = args.steps or (8 if config == 4 else 6)
    warmup = args.warmup if args.warmup >= 0 else 1
    mlds, xs, kw, desc = build_workload(config, B, T, rank * B, args.level1_taps)
    nlev = mlds.getNbLevels()
    x_dev = torch.from_numpy(xs).to(dev)
    hcmp = HierarchicalConvolutionalMatchingPursuit(method=getattr(args, 'method', 'cmp'), device=ctx['dev_index'])

    # Timed step: inputs resident in HBM, and -- as in the config-2 bench -- the residual SAMPLES stay there too: the
    # encoder returns the coefficient matrices and the residual energy of every signal (summed on the device), which is what
    # the reconstruction check needs.  The rate with the float64 residuals crossing PCIe every step (537 MB for 1024
    # signals) is measured separately below (`value_incl_residual_transfer`).
    def step(residuals='energy'):
        return hcmp.computeCoefficientsBatch(xs, mlds, deviceInput=x_dev.data_ptr(), residuals=residuals, **kw)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        coefs, energies, timings = step()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    nsel_local = int(sum(tm['selections'] for tm in timings))
    # the same step with the residual samples fetched (never `value`)
    nres = max(1, min(3, steps))
    t1 = time.perf_counter()
    for _ in range(nres):
        _, residuals, _ = step('samples')
    torch.cuda.synchronize(dev)
    elapsed_res = (time.perf_counter() - t1) / nres
    energies_match = bool(np.allclose(energies, np.sum(np.square(residuals.reshape((B, -1))), axis=1), rtol=1e-10, atol=0.0))
    del residuals
    # SURVEY 8(d)'s wall clock: the signals cross PCIe into the encoder every step (host array in, no device pointer) and the
    # per-signal results (coefficient matrices, residual energies) come back -- never `value`
    nin = max(1, min(3, steps))
    t2 = time.perf_counter()
    for _ in range(nin):
        hcmp.computeCoefficientsBatch(xs, mlds, residuals='energy', **kw)
    torch.cuda.synchronize(dev)
    elapsed_in = (time.perf_counter() - t2) / nin

    # output check of the timed workload: the multilevel code reconstructs the signals
    snr = 10 * np.log10(np.sum(xs.astype(np.float64) ** 2, axis=1) / np.maximum(energies, 1e-300))
    # (what a hierarchy reconstructs at is below its first level's target: 5 dB covers the greedy method on these workloads, the
    #  re-fitted codes of LoCOMP -- fewer, larger coefficients -- end a fraction of a dB lower at config 5)
    floor = kw['toleranceSnr'][0] - (5.0 if getattr(args, 'method', 'cmp') == 'cmp' else 6.0)
    consistent = not (config == 4 and args.level1_taps == 16)
    check = {'snr_db_min': float(snr.min()), 'snr_db_median': float(np.median(snr)), 'snr_floor_db': floor,
             'reconstructs': bool(snr.min() >= floor) if consistent else None,
             'device_energies_match_fetched_residuals': energies_match,
             'nnz_per_level_mean': [float(np.mean([c[l].nnz for c in coefs])) for l in range(nlev)]}
    # (a failed check is reported in the line, after the collectives below: a rank that left here would hang the others)
    if (consistent and snr.min() < floor) or not energies_match:
        check['FAILED'] = True

    import bench as _b
    # ---- gather of the per-signal multilevel results over the ranks (north_star: "gather of per-signal results only"; configs[4] is
    #      "batch sharded 8 GPU"): event records (hsc/dataset.py:798-811) + float64 values + counts + residual energies, rebuilt
    #      into per-level matrices on every rank.  After the timed region, reported beside the metric.
    gather = None
    if world > 1:
        try:
            from hsc_amd import parallel
            coefs_g, energies_g, _, events_g = hcmp.computeCoefficientsBatch(xs, mlds, deviceInput=x_dev.data_ptr(), residuals='energy',
                                                                             returnEvents=True, **kw)
            vals = [parallel._values_of_events(c, e) for c, e in zip(coefs_g, events_g)]
            fence()
            t3 = time.perf_counter()
            g = parallel.gather_hierarchical(events_g, energies_g, vals, device=dev if args.backend == 'nccl' else None)
            if args.backend == 'nccl':
                torch.cuda.synchronize(dev)
            t_coll = time.perf_counter() - t3
            counts = [int(mlds.getRawDictionary(l).shape[0]) for l in range(nlev)]
            mats = [parallel.events_to_level_matrices(e, counts, T, g['values64'][i]) for i, e in enumerate(g['events'])]
            mine = all((mats[rank * B + b][l] != coefs_g[b][l]).nnz == 0 for b in range(0, B, max(1, B // 8)) for l in range(nlev))
            gather = {'ms': 1e3 * t_coll, 'rebuild_ms': 1e3 * (time.perf_counter() - t3 - t_coll), 'bytes_per_signal': g['bytes_per_signal'],
                      'bytes_this_rank': g['bytes_total'], 'signals': len(g['events']), 'own_shard_matches_encoder': bool(mine)}
            del coefs_g, events_g, vals, g, mats
        except Exception as ex:            # (reported beside the metric, never part of it)
            gather = {'error': '%s: %s' % (type(ex).__name__, ex)}
    elapsed_max, nsel_total = _b.reduce_over_ranks(ctx, args, elapsed, nsel_local)
    elapsed_in_max, _ = _b.reduce_over_ranks(ctx, args, elapsed_in, 0)
    if rank != 0:
        return None
    kernel_ms = float(sum(sum(tm['kernel_ms'][:3]) for tm in timings))
    D0 = mlds.getRawDictionary(0)
    K0, W0 = D0.shape[0], D0.shape[1]
    levels = []
    for l, tm in enumerate(timings):
        Dl = mlds.getRawDictionary(l)
        e = {'level': l, 'dictionary': list(Dl.shape), 'variant': tm['variant'], 'selections': tm['selections'], 'stop_reasons': tm.get('stops'),
             'prepare_ms': tm['kernel_ms'][0], 'init_ms': tm['kernel_ms'][1], 'loop_ms': tm['kernel_ms'][2],
             'selections_per_s': tm['selections'] / (1e-3 * max(sum(tm['kernel_ms'][:3]), 1e-9))}
        if l == 0:
            fl_init = 2.0 * T * K0 * W0 * B
            fl_loop = 2.0 * (2 * W0 - 1) * K0 * W0 * tm['selections']
            e.update(bound='mfma', init_tflops=fl_init / (tm['kernel_ms'][1] * 1e-3) / 1e12, loop_tflops=fl_loop / (tm['kernel_ms'][2] * 1e-3) / 1e12,
                     loop_frac=fl_loop / (tm['kernel_ms'][2] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
        else:
            e.update(level_roofline(Dl, T, B, tm, nbBlocks=kw['nbBlocks']))
        levels.append(e)
    # PMC traffic figures only from a summary collected on THIS tree's kernels (tools/csrc_digest.py stamps it)
    from tools_csrc_digest import load_pmc_summary
    pmc, pmc_stale = load_pmc_summary(os.path.join(ROOT, 'profiles', 'pmc_summary_hsc%d%s.json' % (config, '_locomp' if getattr(args, 'method', 'cmp') == 'locomp' else '')))
    if pmc_stale:
        pmc = None
    dom = max(levels, key=lambda e: e['loop_ms'] + e['init_ms'])
    l0 = levels[0]
    out = {
        'metric': 'atom-selections/sec (+ residual-energy match) on 1-D CSC',
        'value': nsel_total * steps / elapsed_max, 'unit': 'atom-selections/s',
        'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * elapsed_max / steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 (level 0) / f64 (levels >= 1)', 'data': 'synthetic',
        'config': {'workload': desc, 'method': getattr(args, 'method', 'cmp'), 'signals_per_gpu': B, 'T': T, 'levels': nlev, 'selections_per_step': nsel_total,
                   'kernel_ms_per_step': kernel_ms, 'output_check': check},
        'roofline': {'bound': 'mfma', 'kernel': 'level-0 greedy loop (iterate_kernel, blocked selection)', 'achieved': l0['loop_tflops'],
                     'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': l0['loop_frac'], 'kernel_ms': l0['loop_ms'],
                     'traffic': (pmc or {}).get('level0_loop_hbm_bytes_per_launch'), 'dominant_level': dom['level'],
                     'levels': levels, 'pmc': pmc, 'pmc_stale': pmc_stale},
        'value_incl_transfers': nsel_total / elapsed_in_max,
        'transfers': {'ms_per_step': 1e3 * elapsed_in_max, 'h2d_bytes_per_step': int(xs.nbytes),
                      'note': 'the same step fed from a (pageable) host array: H2D of the signals, the encode, D2H of the coefficient matrices and residual energies'},
        'gather': gather,
        'value_incl_residual_transfer': nsel_local * world / elapsed_res,
        'residual_transfer': {'ms_per_step': 1e3 * elapsed_res, 'd2h_bytes_per_step': int(B * T * 8),
                              'note': 'the same step with the float64 residual samples of every signal fetched to the host (this rank)'},
        'cpu_baseline': None,
    }
    if world == 1 and not args.no_cpu_baseline:
        nproc = args.cpu_procs or min(16, _b.host_cores())
        payload = json.dumps(dict(config=config, nproc=nproc, T=T, level1_taps=args.level1_taps))
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), payload], env=env, capture_output=True, text=True, timeout=1500)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
            out['cpu_baseline'] = json.loads(line[-1]) if line else {'error': (res.stderr or 'no output')[-400:]}
        except Exception as ex:
            out['cpu_baseline'] = {'error': str(ex)}
    hcmp.close()
    if check.get('FAILED'):
        out['value'] = None                 # a broken encode must not leave a valid-looking throughput figure behind
        out['failed'] = 'output_check'
    return out


def compact(out):
    """The figures of a hierarchical bench line that bench.py's default run carries in its `secondary` object."""
    if out is None:
        return None
    keep = {k: out[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'value_incl_transfers', 'value_incl_residual_transfer')}
    keep['workload'] = out['config']['workload']
    keep['method'] = out['config'].get('method', 'cmp')
    keep['signals_per_gpu'] = out['config']['signals_per_gpu']
    keep['selections_per_step'] = out['config']['selections_per_step']
    keep['kernel_ms_per_step'] = out['config']['kernel_ms_per_step']
    keep['output_check'] = out['config']['output_check']
    keep['residual_transfer_ms_per_step'] = out['residual_transfer']['ms_per_step']
    keep['levels'] = out['roofline']['levels']
    return keep


if __name__ == '__main__':
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    cpu_baseline_main(sys.argv[1])
