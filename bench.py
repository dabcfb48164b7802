#!/usr/bin/env python
"""bench.py -- atom-selections/sec of the convolutional matching-pursuit hot path on MI355X.

One "step" = one full encode of a batch of independent 1-D signals (BASELINE.json config 2:
B=1024 signals of T=65536 samples per GPU, 256-atom x 64-tap dictionary, L0=256, float32,
nbBlocks=1): prepare + initial correlation (hsc/modeling.py:1077) + 256 greedy
select/subtract/re-correlate iterations per signal (:1086-1163), inputs already resident in HBM.
Signals shard across GPUs with no data-path collective (each rank generates and encodes its own
B signals => weak scaling); the only collectives are the timing barrier / max / sum.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline      the dominant kernel's algorithmic FLOP / its HIP-event duration vs the fp32 MFMA peak
  cpu_baseline  the NumPy port of the reference's CPU path (oracle/numpy_port.py) timed on this
                box's host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = fp32 vector peak
PEAK_FP64_MFMA_TFLOPS = 78.6      # AMD MI355X datasheet (v_mfma_f64_16x16x4_f64); used with --dtype f64 only


def cpu_baseline(cfg, nproc, per_proc):
    """Time the NumPy port on `nproc` processes x `per_proc` signals of the bench workload.
    Runs BEFORE anything touches the GPU (fork-based pool)."""
    import multiprocessing as mp
    import numpy as np
    import hsc_amd.synth as synth
    from oracle import numpy_port as port

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
    return {
        'value': nsel / wall, 'unit': 'atom-selections/s', 'cores': nproc, 'kind': 'port',
        'sample': '%d signals (%d processes x %d) of the bench workload, NumPy port of hsc/modeling.py:1053-1186 '
                  '(oracle/numpy_port.py), BLAS threads=1 per process, wall %.1f s' % (nproc * per_proc, nproc, per_proc, wall),
        'per_core': nsel / wall / nproc,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=1024, help='signals per GPU')
    ap.add_argument('--T', type=int, default=65536)
    ap.add_argument('--K', type=int, default=256)
    ap.add_argument('--W', type=int, default=64)
    ap.add_argument('--L0', type=int, default=256)
    ap.add_argument('--kind', default='planted', choices=['planted', 'noise'])
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'], help='arithmetic type of the path (BASELINE config 2 is f32)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: 'nccl' (= RCCL over xGMI) or 'gloo' (rehearsal)")
    ap.add_argument('--cpu-procs', type=int, default=0, help='processes of the CPU baseline (0 = min(8, cores))')
    ap.add_argument('--cpu-signals-per-proc', type=int, default=4, help='signals each CPU process encodes (~3 s each)')
    ap.add_argument('--profile-steps', type=int, default=3, help='extra untimed steps used for per-kernel HIP-event timing')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                             % (args.gpus, args.gpus))
        raise SystemExit('WORLD_SIZE=%d does not match --gpus %d' % (world, args.gpus))
    cfg = dict(B=args.batch, T=args.T, K=args.K, W=args.W, L0=args.L0, kind=args.kind, dtype=args.dtype)

    os.environ.setdefault('OMP_NUM_THREADS', '1')
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nproc = args.cpu_procs or min(8, os.cpu_count() or 1)
        cpu = cpu_baseline(cfg, nproc, args.cpu_signals_per_proc)

    import numpy as np
    import torch
    import torch.distributed as dist
    import hsc_amd.synth as synth
    from hsc_amd import _native

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

    # ---- synthetic inputs: this rank's shard of the (weakly scaled) batch, resident in HBM
    npdt = np.float64 if args.dtype == 'f64' else np.float32
    D = synth.make_dictionary(cfg['K'], cfg['W'], seed=2, dtype=npdt)
    first = rank * cfg['B']
    x_host = synth.make_batch(D, cfg['T'], first, cfg['B'], kind=cfg['kind'], nb_atoms=cfg['L0'], seed=2, dtype=npdt)
    x = torch.from_numpy(x_host).to(dev)
    del x_host

    stream = torch.cuda.Stream(device=dev)
    eng = _native.Engine(dev_index)
    eng.set_stream(stream.cuda_stream)
    eng.set_dictionary(D)
    params = _native.make_params(nbNonzeroCoefs=cfg['L0'], nbBlocks=1, minCoefficients=1e-16,
                                 eps=float(np.finfo(npdt).eps), maxEvents=2 * cfg['L0'] + 64)

    def step():
        eng.encode_batch_device(x.data_ptr(), cfg['B'], cfg['T'], params)

    def fence():
        stream.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
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

    # ---- per-kernel durations (HIP events on the engine's stream), untimed extra steps
    kms = np.zeros(4, dtype=np.float64)
    for _ in range(max(1, args.profile_steps)):
        step()
        stream.synchronize()
        kms += eng.last_kernel_ms().astype(np.float64)
    kms /= max(1, args.profile_steps)

    cdev = dev if args.backend == 'nccl' else torch.device('cpu')
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    t_sel = torch.tensor([float(nsel_local)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_sel, op=dist.ReduceOp.SUM)
    elapsed_max = float(t_el.item())
    nsel_total = float(t_sel.item())

    if rank == 0:
        F = 1
        flop_init = 2.0 * cfg['T'] * cfg['K'] * cfg['W'] * F * cfg['B']                       # per launch
        flop_loop = 2.0 * (2 * cfg['W'] - 1) * cfg['K'] * cfg['W'] * F * nsel_local           # per launch
        kern = [('corr_init (initial correlation, modeling.py:1077)', flop_init, kms[1]),
                ('iterate (greedy loop re-correlation, modeling.py:1018-1051)', flop_loop, kms[2])]
        dom = max(kern, key=lambda k: k[2])
        achieved = dom[1] / (dom[2] * 1e-3) / 1e12
        peak = PEAK_FP64_MFMA_TFLOPS if args.dtype == 'f64' else PEAK_FP32_MFMA_TFLOPS
        # HBM bytes per launch of the dominant kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate runs, gfx950 correction applied by tools/parse_pmc.py); null if not collected
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_summary.json')))
            key = 'corr_init_mfma_kernel' if dom is kern[0] else 'iterate_kernel'
            if cfg['B'] == 1024 and cfg['T'] == 65536 and variant.startswith('mfma') and args.dtype == 'f32':
                traffic = pmc[key]['hbm_bytes_per_launch']
        except Exception:
            traffic = None
        out = {
            'metric': 'atom-selections/sec (+ residual-energy match) on 1-D CSC',
            'value': nsel_total * args.steps / elapsed_max,
            'unit': 'atom-selections/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed_max / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: single-level CSC, %d signals/GPU x len %d, %d-atom x %d-tap dict, '
                                   'L0=%d, nbBlocks=1, %s signals' % (cfg['B'], cfg['T'], cfg['K'], cfg['W'], cfg['L0'], cfg['kind']),
                       'signals_per_gpu': cfg['B'], 'T': cfg['T'], 'K': cfg['K'], 'W': cfg['W'], 'L0': cfg['L0'],
                       'selections_per_step': nsel_total, 'variant': variant,
                       'stop_reasons': {_native.STOP_NAMES[i]: int(n) for i, n in enumerate(stops) if n}},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_unit': 'HBM bytes/launch (PMC)', 'kernel': dom[0],
                         'kernel_ms': float(dom[2]),
                         'all_kernels': [{'kernel': k[0], 'algorithmic_tflop': k[1] / 1e12, 'ms': float(k[2]),
                                          'tflops': (k[1] / (k[2] * 1e-3) / 1e12) if k[2] > 0 else None} for k in kern],
                         'prepare_ms': float(kms[0]),
                         'whole_job_tflops': (flop_init + flop_loop) * args.steps * world / elapsed_max / 1e12},
            'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
