#!/usr/bin/env python
"""bench.py -- atom-selections/sec of the convolutional matching-pursuit hot path on MI355X.

--config 2 (default, the configuration BASELINE.json's metric is quoted on)
    One "step" = one full encode of a batch of independent 1-D signals (B=1024 signals of T=65536 samples per GPU,
    256-atom x 64-tap dictionary, L0=256, float32, nbBlocks=1): prepare + initial correlation (hsc/modeling.py:1077) +
    256 greedy select/subtract/re-correlate iterations per signal (:1086-1163), inputs already resident in HBM.
--config 4 | 5 (hierarchical encoder, hsc/modeling.py:1427-1654; see bench_hsc below)
    One step = one multilevel encode of the batch: level 0 on the signals, every further level on the previous level's
    coefficient streams (device-chained), then the device epilogue (redistribution, CSC, events, residual).

Signals shard across GPUs with no data-path collective (each rank generates and encodes its own B signals => weak
scaling); the collectives are the timing barrier / max / sum and, after the timed region, the gather of the per-signal
results (hsc_amd.parallel.gather_results: fixed-shape tensors through all_gather_into_tensor, reported separately).

  python bench.py --gpus 1 --steps 200 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), with:
  roofline      dominant kernel: algorithmic FLOP / its HIP-event duration vs the fp32 MFMA peak; plus the HBM side of
                SURVEY 8(d): hbm_fraction (PMC bytes/s / 8 TB/s) and table_scan_equiv (what the reference's formulation --
                re-scan the [T,K] table for every selection -- would need in HBM bandwidth at the measured rate)
  value_incl_transfers   the same encode fed from pinned host memory (H2D of step i+1 under the kernels of step i) with the
                per-signal results (events, stats, energies) copied back every step
  cpu_baseline  the NumPy port of the reference's CPU path (oracle/numpy_port.py) timed AFTER the GPU section in a fresh
                child process (one worker per host core) on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = fp32 vector peak
PEAK_FP64_MFMA_TFLOPS = 78.6      # AMD MI355X datasheet (v_mfma_f64_16x16x4_f64); used with --dtype f64 only
PEAK_HBM_BYTES = 8.0e12           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def _spin(n):
    t0 = time.perf_counter()
    x = 0
    for i in range(n):
        x += i * i
    return time.perf_counter() - t0


def measured_usable_cores(limit=64):
    """How many cores the box really gives this job: N spinning processes against one (the GPU boxes show 256 logical cores
    to os.cpu_count() and to the affinity mask, but schedule about one GPU's share of them)."""
    import multiprocessing as mp
    n = min(limit, host_cores())
    if n <= 1:
        return 1
    work = 5000000
    t1 = min(_spin(work) for _ in range(2))
    with mp.get_context('fork').Pool(n) as pool:
        pool.map(_spin, [1000] * n)                      # (workers up before the clock starts)
        wall = []
        for _ in range(2):
            t0 = time.perf_counter()
            pool.map(_spin, [work] * n, chunksize=1)
            wall.append(time.perf_counter() - t0)
    return max(1, min(n, int(round(n * t1 / min(wall)))))


def cpu_baseline_main(args):
    """Child-process entry (--cpu-baseline-only): never touches the GPU.  Times the NumPy port on one worker per core."""
    import multiprocessing as mp
    import numpy as np
    import hsc_amd.synth as synth
    from oracle import numpy_port as port

    cfg = json.loads(args.cpu_baseline_only)
    nproc, per_proc = cfg['nproc'], cfg['per_proc']
    usable = measured_usable_cores()
    npdt = np.float64 if cfg.get('dtype') == 'f64' else np.float32
    D = synth.make_dictionary(cfg['K'], cfg['W'], seed=2, dtype=npdt)
    jobs = []
    for p in range(nproc):
        sig = [synth.make_signal(D, cfg['T'], p * per_proc + i, kind=cfg['kind'], nb_atoms=cfg['L0'], seed=2, dtype=npdt)
               for i in range(per_proc)]
        jobs.append((D, sig, cfg['L0']))
    t0 = time.perf_counter()
    if nproc == 1:
        out = [port._timed_worker(jobs[0])]
    else:
        with mp.get_context('fork').Pool(nproc) as pool:
            out = pool.map(port._timed_worker, jobs)
    wall = time.perf_counter() - t0
    nsel = int(sum(o[0] for o in out))
    print(json.dumps({
        'value': nsel / wall, 'unit': 'atom-selections/s', 'cores': nproc, 'kind': 'port',
        'host_cores': os.cpu_count(), 'affinity_cores': host_cores(), 'usable_cores': usable,
        'usable_cores_note': 'measured: speed-up of min(64, affinity) spinning processes over one',
        'sample': '%d signals (%d processes x %d) of the bench workload, NumPy port of hsc/modeling.py:1053-1186 '
                  '(oracle/numpy_port.py), BLAS threads=1 per process, wall %.1f s' % (nproc * per_proc, nproc, per_proc, wall),
        'per_core': nsel / wall / nproc}))


def run_cpu_baseline(cfg, nproc, per_proc):
    """Fresh child process (this one has initialised the GPU: it is neither forked nor re-executed)."""
    payload = dict(cfg, nproc=nproc, per_proc=per_proc)
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', json.dumps(payload)],
                             env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        return json.loads(line[-1]) if line else {'error': (out.stderr or 'no output')[-400:]}
    except Exception as ex:                       # the GPU numbers stand on their own
        return {'error': str(ex)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=0, help='timed steps (0 = default of the configuration: >= 5 s of GPU time)')
    ap.add_argument('--warmup', type=int, default=-1)
    ap.add_argument('--config', type=int, default=2, choices=[2, 4, 5], help='BASELINE.json configs[] number (2 = headline, 4 / 5 = hierarchical)')
    ap.add_argument('--batch', type=int, default=0, help='signals per GPU (0 = the configuration\'s)')
    ap.add_argument('--T', type=int, default=65536)
    ap.add_argument('--K', type=int, default=256)
    ap.add_argument('--W', type=int, default=64)
    ap.add_argument('--L0', type=int, default=256)
    ap.add_argument('--kind', default='planted', choices=['planted', 'noise'])
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'], help='arithmetic type of the path (BASELINE config 2 is f32)')
    ap.add_argument('--level1-taps', type=int, default=17, help='config 4: taps of the level-1 dictionary (17 = scales [64, 80], the '
                    'nearest shape whose hierarchy is self-consistent; 16 = the literal BASELINE shape, see hsc_amd.synth.make_hierarchy)')
    ap.add_argument('--method', default='cmp', choices=['cmp', 'locomp'], help="configs 4 / 5: method of the hierarchical encoder ('locomp' is the "
                    "reference's default, modeling.py:1429; BASELINE's metric is quoted on the greedy loop, 'cmp')")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-transfers', action='store_true', help='skip the PCIe-inclusive leg')
    ap.add_argument('--no-secondary', action='store_true', help='config 2 only: skip the bounded runs of the hierarchical configurations (configs 4 / 5)')
    ap.add_argument('--secondary-budget-s', type=int, default=150, help='wall-clock budget of that section')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: 'nccl' (= RCCL over xGMI) or 'gloo' (rehearsal)")
    ap.add_argument('--cpu-procs', type=int, default=0, help='processes of the CPU baseline (0 = one per usable core, at most 16: the CPU share of one GPU; the port is memory-bound and slows down beyond that)')
    ap.add_argument('--cpu-signals-per-proc', type=int, default=2, help='signals each CPU process encodes (~3-6 s each)')
    ap.add_argument('--profile-steps', type=int, default=5, help='extra untimed steps used for per-kernel HIP-event timing')
    ap.add_argument('--cpu-baseline-only', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_baseline_main(args)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                             % (args.gpus, args.gpus))
        raise SystemExit('WORLD_SIZE=%d does not match --gpus %d' % (world, args.gpus))
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no GPU visible); there is no CPU path to fall back to')
    dev_index = local_rank % torch.cuda.device_count()     # one rank per GPU; wraps only in 1-GPU rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)     # RCCL
        else:
            dist.init_process_group(args.backend)
    ctx = dict(rank=rank, world=world, dev=dev, dev_index=dev_index, dist=dist, torch=torch)
    if args.config == 2:
        out = bench_cmp(args, ctx)
    else:
        import bench_hsc
        out = bench_hsc.run(args, ctx)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and out is not None and (out.get('failed') or out.get('config', {}).get('output_check', {}).get('FAILED')):
        sys.exit(3)                         # the line is out (ranks never hang on it), but a failed output check is a failed run


def reduce_over_ranks(ctx, args, elapsed, nsel):
    torch, dist = ctx['torch'], ctx['dist']
    cdev = ctx['dev'] if args.backend == 'nccl' else torch.device('cpu')
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    t_sel = torch.tensor([float(nsel)], dtype=torch.float64, device=cdev)
    if ctx['world'] > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_sel, op=dist.ReduceOp.SUM)
    return float(t_el.item()), float(t_sel.item())


def bench_cmp(args, ctx):
    import numpy as np
    import hsc_amd.synth as synth
    from hsc_amd import _native, parallel
    torch, dist, dev, rank, world = ctx['torch'], ctx['dist'], ctx['dev'], ctx['rank'], ctx['world']
    cfg = dict(B=args.batch or 1024, T=args.T, K=args.K, W=args.W, L0=args.L0, kind=args.kind, dtype=args.dtype)
    steps = args.steps or 200            # ~25 ms each: >= 5 s of GPU time by default
    warmup = args.warmup if args.warmup >= 0 else 5

    # ---- synthetic inputs: this rank's shard of the (weakly scaled) batch, resident in HBM
    npdt = np.float64 if args.dtype == 'f64' else np.float32
    D = synth.make_dictionary(cfg['K'], cfg['W'], seed=2, dtype=npdt)
    first = rank * cfg['B']
    x_host = synth.make_batch(D, cfg['T'], first, cfg['B'], kind=cfg['kind'], nb_atoms=cfg['L0'], seed=2, dtype=npdt)
    x = torch.from_numpy(x_host).to(dev)

    stream = torch.cuda.Stream(device=dev)
    eng = _native.Engine(ctx['dev_index'])
    eng.set_stream(stream.cuda_stream)
    eng.set_dictionary(D)
    params = _native.make_params(nbNonzeroCoefs=cfg['L0'], nbBlocks=1, minCoefficients=1e-16,
                                 eps=float(np.finfo(npdt).eps), maxEvents=2 * cfg['L0'] + 64)

    def step(ptr=None):
        eng.encode_batch_device(ptr or x.data_ptr(), cfg['B'], cfg['T'], params)

    def fence():
        stream.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    stream.synchronize()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()

    stats = eng.fetch_stats()
    nsel_local = int(stats[:, _native.STAT_ITERATIONS].sum())
    stops = np.bincount(stats[:, _native.STAT_STOP], minlength=8)
    variant = eng.last_variant()
    # output check of the timed workload itself (cheap, every run): stop rule, counts, tracked energy vs the residual
    energies = eng.fetch_energies()
    check = {'all_stop_nnz': bool(np.all(stats[:, _native.STAT_STOP] == 2)),
             'all_nnz_eq_L0': bool(np.all(stats[:, _native.STAT_NNZ] == cfg['L0'])),
             'energy_decreased': bool(np.all(energies[:, 1] < energies[:, 0]))}
    r = eng.fetch_residual()
    e_rec = np.sum(np.square(r.astype(np.float64)), axis=(1, 2))
    check['tracked_energy_matches_residual'] = bool(np.all(np.abs(e_rec - energies[:, 1]) <= 1e-5 * energies[:, 0]))
    del r

    # ---- per-kernel durations (HIP events on the engine's stream), untimed extra steps
    #      in the regime of the timed loop: three encodes queued back to back, the events of the last one read, repeated
    kms = np.zeros(4, dtype=np.float64)
    for _ in range(max(1, args.profile_steps)):
        for _ in range(3):
            step()
        stream.synchronize()
        kms += eng.last_kernel_ms().astype(np.float64)
    kms /= max(1, args.profile_steps)

    # ---- the same encode fed over PCIe: pinned host buffers, H2D of step i+1 under the kernels of step i, the
    #      per-signal results (events in selection order, stats, energies) back to pinned memory every step
    incl = None
    if not args.no_transfers:
        nt = max(4, min(steps, 40))
        x_pin = torch.from_numpy(x_host).pin_memory()
        xd = [torch.empty_like(x), torch.empty_like(x)]
        cap = int(params.max_events)
        ev_t = torch.empty((cfg['B'], cap), dtype=torch.int32).pin_memory()
        ev_k = torch.empty((cfg['B'], cap), dtype=torch.int32).pin_memory()
        ev_c = torch.empty((cfg['B'], cap), dtype=torch.float64 if args.dtype == 'f64' else torch.float32).pin_memory()
        st_pin = torch.empty((cfg['B'], _native.STAT_COUNT), dtype=torch.int32).pin_memory()
        en_pin = torch.empty((cfg['B'], 2), dtype=torch.float64).pin_memory()
        copy_stream = torch.cuda.Stream(device=dev)
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        freed = [torch.cuda.Event(), torch.cuda.Event()]

        def upload(i):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[i & 1])                 # the encode that read this buffer has finished
                xd[i & 1].copy_(x_pin, non_blocking=True)
                ready[i & 1].record(copy_stream)

        for e in freed:
            e.record(stream)
        fence()
        upload(0)
        t1 = time.perf_counter()
        for i in range(nt):
            if i + 1 < nt:
                upload(i + 1)
            stream.wait_event(ready[i & 1])
            step(xd[i & 1].data_ptr())
            freed[i & 1].record(stream)
            eng.fetch_events_into(ev_t.data_ptr(), ev_k.data_ptr(), ev_c.data_ptr())       # D2H on the engine's stream + sync
            eng.fetch_stats_into(st_pin.data_ptr())
            eng.fetch_energies_into(en_pin.data_ptr())
        stream.synchronize()
        torch.cuda.synchronize(dev)
        el_t = time.perf_counter() - t1
        if world > 1:
            dist.barrier()
        el_t_max, _ = reduce_over_ranks(ctx, args, el_t, 0)
        incl = {'steps': nt, 'ms_per_step': 1e3 * el_t_max / nt,
                'h2d_bytes_per_step': int(x_pin.numel() * x_pin.element_size()),
                'd2h_bytes_per_step': int(ev_t.numel() * 4 * 2 + ev_c.numel() * ev_c.element_size() + st_pin.numel() * 4 + en_pin.numel() * 8),
                'note': 'pinned host buffers; input H2D of step i+1 overlaps the kernels of step i; events / stats / energies D2H every step; residuals stay on the device'}
    del x_host

    # ---- gather of the per-signal results over the ranks (north_star: "scatter/gather of per-signal results only")
    gather = None
    if world > 1:
        t2 = time.perf_counter()
        try:                              # (reported beside the metric, never part of it: a failure here must not cost the line)
            g = parallel.gather_results(eng, device=dev if args.backend == 'nccl' else None)
            if args.backend == 'nccl':
                torch.cuda.synchronize(dev)
            gather = {'ms': 1e3 * (time.perf_counter() - t2), 'bytes_per_signal': g['bytes_per_signal'], 'signals': int(g['stats'].shape[0])}
        except Exception as ex:
            gather = {'error': '%s: %s' % (type(ex).__name__, ex)}

    elapsed_max, nsel_total = reduce_over_ranks(ctx, args, elapsed, nsel_local)
    if rank != 0:
        return None

    F = 1
    flop_init = 2.0 * cfg['T'] * cfg['K'] * cfg['W'] * F * cfg['B']                       # per launch
    flop_loop = 2.0 * (2 * cfg['W'] - 1) * cfg['K'] * cfg['W'] * F * nsel_local           # per launch
    kern = [('corr_init (initial correlation, modeling.py:1077)', flop_init, kms[1], 'corr_init_mfma_kernel'),
            ('iterate (greedy loop re-correlation, modeling.py:1018-1051)', flop_loop, kms[2], 'iterate_kernel')]
    dom = max(kern, key=lambda k: k[2])
    achieved = dom[1] / (dom[2] * 1e-3) / 1e12
    peak = PEAK_FP64_MFMA_TFLOPS if args.dtype == 'f64' else PEAK_FP32_MFMA_TFLOPS
    # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction
    # applied by tools/parse_pmc.py; profiles/pmc_summary.json names the build they were collected on); null if the
    # summary does not cover this shape
    traffic, hbm_fraction, pmc_src, pmc_stale = None, None, None, None
    try:
        from tools_csrc_digest import load_pmc_summary
        pmc, pmc_stale = load_pmc_summary(os.path.join(ROOT, 'profiles', 'pmc_summary.json'))
        # (a summary collected on other kernels than this tree's is not reported: traffic = null, pmc_stale = true)
        if pmc is not None and not pmc_stale and cfg['B'] == 1024 and cfg['T'] == 65536 and variant.startswith('mfma') and args.dtype == 'f32':
            traffic = pmc[dom[3]]['hbm_bytes_per_launch']
            step_bytes = sum(pmc[k]['hbm_bytes_per_launch'] for k in ('prepare_kernel', 'corr_init_mfma_kernel', 'iterate_kernel') if k in pmc)
            hbm_fraction = step_bytes / (elapsed_max / steps) / PEAK_HBM_BYTES
            pmc_src = pmc.get('_source')
    except Exception:
        traffic = None
    rate = nsel_total * steps / elapsed_max
    esz = 8 if args.dtype == 'f64' else 4
    table_bytes = cfg['T'] * cfg['K'] * esz                                                # one scan of the [T,K] table per selection (:967)
    out = {
        'metric': 'atom-selections/sec (+ residual-energy match) on 1-D CSC',
        'value': rate,
        'unit': 'atom-selections/s',
        'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': 1e3 * elapsed_max / steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: single-level CSC, %d signals/GPU x len %d, %d-atom x %d-tap dict, '
                               'L0=%d, nbBlocks=1, %s signals' % (cfg['B'], cfg['T'], cfg['K'], cfg['W'], cfg['L0'], cfg['kind']),
                   'signals_per_gpu': cfg['B'], 'T': cfg['T'], 'K': cfg['K'], 'W': cfg['W'], 'L0': cfg['L0'],
                   'selections_per_step': nsel_total, 'variant': variant,
                   'stop_reasons': {_native.STOP_NAMES[i]: int(n) for i, n in enumerate(stops) if n},
                   'output_check': check},
        'value_incl_transfers': (nsel_total * 1e3 / incl['ms_per_step']) if incl else None,
        'transfers': incl,
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': achieved / peak, 'traffic': traffic, 'traffic_unit': 'HBM bytes/launch (PMC)', 'kernel': dom[0],
                     'kernel_ms': float(dom[2]),
                     'all_kernels': [{'kernel': k[0], 'algorithmic_tflop': k[1] / 1e12, 'ms': float(k[2]),
                                      'tflops': (k[1] / (k[2] * 1e-3) / 1e12) if k[2] > 0 else None,
                                      'frac': (k[1] / (k[2] * 1e-3) / 1e12 / peak) if k[2] > 0 else None} for k in kern],
                     'prepare_ms': float(kms[0]),
                     'whole_job_tflops': (flop_init + flop_loop) * steps * world / elapsed_max / 1e12,
                     'whole_job_frac': (flop_init + flop_loop) * steps * world / elapsed_max / 1e12 / peak / world,
                     # the HBM side: measured bytes/s of the table-free formulation against 8 TB/s, and the
                     # bandwidth the reference's formulation (one scan of the [T,K] table per selection) would need at this rate
                     'hbm_fraction': hbm_fraction, 'pmc_source': pmc_src, 'pmc_stale': pmc_stale,
                     'table_scan_equiv': {'bytes_per_selection': table_bytes, 'tb_per_s_needed': rate / world * table_bytes / 1e12,
                                          'x_hbm_peak': rate / world * table_bytes / PEAK_HBM_BYTES}},
        'gather': gather,
        'cpu_baseline': None,
    }
    if check and not all(check.values()):
        out['config']['output_check']['FAILED'] = True
    # ---- the hierarchical configurations of BASELINE.json at their own sizes, a few steps each (bounded: it must never
    #      cost the headline line): configs[3] with 17 and with the literal 16 level-1 taps, configs[4] at its per-GPU share
    if world == 1:
        eng.close()
        del x
        torch.cuda.empty_cache()
    if world == 1 and not args.no_secondary:
        out['secondary'] = run_secondary(args, ctx)
    if world == 1 and not args.no_cpu_baseline:
        nproc = args.cpu_procs or min(16, host_cores())
        out['cpu_baseline'] = run_cpu_baseline(cfg, nproc, args.cpu_signals_per_proc)
    return out


def run_secondary(args, ctx):
    import copy
    import bench_hsc
    sec = {}
    t_begin = time.perf_counter()
    # the headline workload in the reference's DEFAULT dtype (float64 inputs, hsc/modeling.py:1053 computes in the input's dtype):
    # the v_mfma_f64_16x16x4_f64 kernels, a few steps, with their own roofline against the fp64 matrix peak
    try:
        a = copy.copy(args)
        a.dtype, a.steps, a.warmup, a.no_cpu_baseline, a.no_secondary, a.no_transfers, a.profile_steps = 'f64', 3, 1, True, True, True, 2
        t0 = time.perf_counter()
        o = bench_cmp(a, ctx)
        sec['config2_f64'] = {'value': o['value'], 'unit': o['unit'], 'ms_per_step': o['ms_per_step'], 'steps': o['steps'], 'dtype': 'f64',
                              'workload': o['config']['workload'], 'variant': o['config']['variant'], 'output_check': o['config']['output_check'],
                              'roofline': {k: o['roofline'][k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'kernel_ms', 'all_kernels', 'whole_job_frac')},
                              'section_wall_s': time.perf_counter() - t0}
    except (Exception, SystemExit) as ex:
        sec['config2_f64'] = {'error': '%s: %s' % (type(ex).__name__, ex)}
    ctx['torch'].cuda.empty_cache()
    # (config 4 also with the method the reference's own script runs it with -- learn_mlcsc_dataset.py:108 builds the encoder with
    #  its default, LoCOMP: the device loop of csrc/hscmp_locomp.h)
    for name, config, taps, method in (('config4_17taps', 4, 17, 'cmp'), ('config4_16taps', 4, 16, 'cmp'), ('config5', 5, 17, 'cmp'),
                                       ('config4_17taps_locomp', 4, 17, 'locomp'), ('config5_locomp', 5, 17, 'locomp')):
        if time.perf_counter() - t_begin > args.secondary_budget_s:
            sec[name] = {'skipped': 'time budget of the secondary section (%d s) spent' % args.secondary_budget_s}
            continue
        a = copy.copy(args)
        a.config, a.level1_taps, a.batch, a.T, a.method = config, taps, 0, 65536, method
        a.steps, a.warmup, a.no_cpu_baseline = (3 if method == 'cmp' else 2), 1, True
        t0 = time.perf_counter()
        try:
            sec[name] = bench_hsc.compact(bench_hsc.run(a, ctx))
            sec[name]['section_wall_s'] = time.perf_counter() - t0
        except (Exception, SystemExit) as ex:  # (reported, never fatal for the headline; a KeyboardInterrupt still ends the run)
            sec[name] = {'error': '%s: %s' % (type(ex).__name__, ex)}
        ctx['torch'].cuda.empty_cache()
    return sec


if __name__ == '__main__':
    main()
