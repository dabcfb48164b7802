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
    hcmp = HierarchicalConvolutionalMatchingPursuit(method='cmp', device=ctx['dev_index'])

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
    assert np.allclose(energies, np.sum(np.square(residuals.reshape((B, -1))), axis=1), rtol=1e-10, atol=0.0)

    # output check of the timed workload: the multilevel code reconstructs the signals
    snr = 10 * np.log10(np.sum(xs.astype(np.float64) ** 2, axis=1) / np.maximum(energies, 1e-300))
    floor = kw['toleranceSnr'][0] - 5.0
    consistent = not (config == 4 and args.level1_taps == 16)
    check = {'snr_db_min': float(snr.min()), 'snr_db_median': float(np.median(snr)), 'snr_floor_db': floor,
             'reconstructs': bool(snr.min() >= floor) if consistent else None,
             'nnz_per_level_mean': [float(np.mean([c[l].nnz for c in coefs])) for l in range(nlev)]}
    if consistent and snr.min() < floor:
        raise SystemExit('bench_hsc: the encode does not reconstruct its input (min SNR %.2f dB < %.1f dB)' % (snr.min(), floor))

    import bench as _b
    elapsed_max, nsel_total = _b.reduce_over_ranks(ctx, args, elapsed, nsel_local)
    if rank != 0:
        return None
    kernel_ms = float(sum(sum(tm['kernel_ms'][:3]) for tm in timings))
    D0 = mlds.getRawDictionary(0)
    K0, W0 = D0.shape[0], D0.shape[1]
    levels = []
    for l, tm in enumerate(timings):
        Dl = mlds.getRawDictionary(l)
        e = {'level': l, 'dictionary': list(Dl.shape), 'variant': tm['variant'], 'selections': tm['selections'],
             'prepare_ms': tm['kernel_ms'][0], 'init_ms': tm['kernel_ms'][1], 'loop_ms': tm['kernel_ms'][2],
             'selections_per_s': tm['selections'] / (1e-3 * max(sum(tm['kernel_ms'][:3]), 1e-9))}
        if l == 0:
            fl_init = 2.0 * T * K0 * W0 * B
            fl_loop = 2.0 * (2 * W0 - 1) * K0 * W0 * tm['selections']
            e.update(bound='mfma', init_tflops=fl_init / (tm['kernel_ms'][1] * 1e-3) / 1e12, loop_tflops=fl_loop / (tm['kernel_ms'][2] * 1e-3) / 1e12,
                     loop_frac=fl_loop / (tm['kernel_ms'][2] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
        else:
            nz = float(np.count_nonzero(Dl)) / Dl.shape[0]
            e.update(bound='l2-latency', note='sparse x sparse: a few dependent L2 round trips per atom; algorithmic cells per atom = '
                     'non-zeros of the atom (%.1f on average) + the listed cells of its 3W-2 window' % nz,
                     us_per_atom_per_signal=1e3 * tm['kernel_ms'][2] * B / max(tm['selections'], 1))
        levels.append(e)
    pmc = None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_summary_hsc%d.json' % config)))
    except Exception:
        pass
    dom = max(levels, key=lambda e: e['loop_ms'] + e['init_ms'])
    l0 = levels[0]
    out = {
        'metric': 'atom-selections/sec (+ residual-energy match) on 1-D CSC',
        'value': nsel_total * steps / elapsed_max, 'unit': 'atom-selections/s',
        'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * elapsed_max / steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 (level 0) / f64 (levels >= 1)', 'data': 'synthetic',
        'config': {'workload': desc, 'signals_per_gpu': B, 'T': T, 'levels': nlev, 'selections_per_step': nsel_total,
                   'kernel_ms_per_step': kernel_ms, 'output_check': check},
        'roofline': {'bound': 'mfma', 'kernel': 'level-0 greedy loop (iterate_kernel, blocked selection)', 'achieved': l0['loop_tflops'],
                     'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': l0['loop_frac'], 'kernel_ms': l0['loop_ms'],
                     'traffic': (pmc or {}).get('level0_loop_hbm_bytes_per_launch'), 'dominant_level': dom['level'],
                     'levels': levels, 'pmc': pmc},
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
    return out


if __name__ == '__main__':
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    cpu_baseline_main(sys.argv[1])
