#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference, loaded through oracle/ref_loader.py);
the fixtures it writes are data: seeded inputs (or their sha256 when they are regenerated from
hsc_amd.synth) and the reference's outputs.  Re-run with `python tools/make_golden.py`.

Files written:
  tests/golden/cmp_small.npz    -- ~40 small CMP cases (f32/f64, odd/even W, F>1, blocked selection,
                                   weights, every stop rule): inputs + ordered (t,k,c) trace + CSC + residual
  tests/golden/functions.npz    -- convolve1d ('same'/'valid'), _selectBestAtoms, _updateInnerProducts
                                   (reflect padding at both edges) on seeded inputs
  tests/golden/cmp_config.npz   -- BASELINE config 1 (T=4096) and 8 full-size config-2 signals
                                   (T=65536, K=256, W=64, L0=256): outputs + input digests
  tests/golden/locomp_small.npz -- LoCOMP (modeling.py:1191-1425) on small seeded problems
  tests/golden/locomp_hier.npz  -- the hierarchical encoder with its default method='locomp' (config-4 dimensions, a generated
                                   3-level dictionary in float32 and float64) and single-level LoCOMP on rank-deficient groups, each
                                   with the conditioning the reference's pseudo-inverse saw
  tests/golden/synth_small.npz  -- dataset synthesis (hsc/dataset.py:412-796) under fixed numpy seeds: generated
                                   multilevel dictionaries (raw, representations, decompositions), Poisson
                                   events with / without rate scaling, rendered signals
  tests/golden/learn_small.npz  -- convolutional k-means dictionary learner (modeling.py:420-524) under fixed seeds and
                                   its window-assignment step (convolve1d_batch + arg-max) on fixed windows
  tests/golden/hsc_medium.npz   -- 2-level hierarchical encoder on generated data (8192 samples, level-1 dictionary
                                   (24+20) x 33 x 24): per-level coefficients of the reference, inputs by seed
  tests/golden/hsc_config4.npz  -- 2-level hierarchical encoder at the BASELINE config-4 dictionary dimensions ((256+128) x 16|17 x 256),
                                   8192 samples; reconstructSignal's dense (fftconvolve) branch on small seeded cases
  tests/golden/hsc_small.npz    -- 3-level hierarchical encoder (method='cmp'): dictionaries with
                                   singleton bases, representations, per-level coefficients, residual
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from refutil import run_reference_cmp, load_reference  # noqa: E402
import hsc_amd.synth as synth  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def small_cases():
    """(name, x, D, kwargs) tuples; all inputs from one seeded RandomState, in a fixed order."""
    rs = np.random.RandomState(20261003)
    norm = load_reference().utils.normalize
    cases = []
    for dt in (np.float32, np.float64):
        tag = 'f32' if dt == np.float32 else 'f64'
        # reference test shapes (tests/hsc/test_modeling.py:247-270): tiny T, odd/even W
        for K in (1, 2, 3):
            for W in (3, 5, 6):
                D = norm(rs.random_sample((K, W)).astype(dt), axis=1)
                x = rs.random_sample(16).astype(dt)
                cases.append(('%s_T16_K%d_W%d_L4' % (tag, K, W), x, D, dict(nbNonzeroCoefs=4)))
        D = norm(rs.random_sample((16, 15, 7)).astype(dt), axis=(1, 2))
        x = rs.random_sample((64, 7)).astype(dt)
        cases.append(('%s_T64_K16_W15_F7_L16' % tag, x, D, dict(nbNonzeroCoefs=16)))
        # stop rules (test_modeling.py:329-360)
        D = norm(rs.random_sample((32, 9)).astype(dt), axis=1)
        x = rs.random_sample(128).astype(dt)
        for tol in (0.5, 0.1, 0.001):
            cases.append(('%s_T128_K32_W9_scale%g' % (tag, tol), x, D, dict(toleranceResidualScale=tol)))
        for tol in (5, 20, 50):
            cases.append(('%s_T128_K32_W9_snr%g' % (tag, tol), x, D, dict(toleranceSnr=tol)))
        # features (test_modeling.py:362-377)
        for F in (4, 11):
            D3 = norm(rs.random_sample((32, 9, F)).astype(dt), axis=(1, 2))
            x3 = rs.random_sample((128, F)).astype(dt)
            cases.append(('%s_T128_K32_W9_F%d_scale0.01' % (tag, F), x3, D3, dict(toleranceResidualScale=0.01)))
        # blocked selection, offsets, weights, interference / weak filters
        D = norm(rs.standard_normal((24, 10)).astype(dt), axis=1)
        x = rs.standard_normal(300).astype(dt)
        for nb in (2, 4, 5, 8, 'auto'):
            cases.append(('%s_T300_K24_W10_snr12_nb%s' % (tag, nb), x, D, dict(toleranceSnr=12, nbBlocks=nb)))
        w = np.ones(24, dtype=dt)
        w[:6] = 0.5
        cases.append(('%s_T300_K24_W10_L30_weights' % tag, x, D, dict(nbNonzeroCoefs=30, weights=w)))
        cases.append(('%s_T300_K24_W10_L30_nb4_weights' % tag, x, D, dict(nbNonzeroCoefs=30, nbBlocks=4, weights=w)))
        cases.append(('%s_T300_K24_W10_snr15_auto_weights' % tag, x, D, dict(toleranceSnr=15, nbBlocks='auto', weights=w)))
        # planted atoms (test_modeling.py:379-396 shape), minCoefficients clip
        D = norm(rs.random_sample((4, 32)).astype(dt), axis=1)
        x = np.zeros(256, dtype=np.float64)
        for c, p, k in zip([1.0, 1.0, 0.5, 1.0, 0.75, 2.0], [32, 48, 64, 96, 128, 192], [0, 3, 1, 0, 2, 2]):
            s, e, es, ee = synth.centered_span(256, 32, p)
            x[s:e] += c * D[k][es:ee]
        cases.append(('%s_planted_T256_K4_W32' % tag, x.astype(dt), D, dict(nbNonzeroCoefs=8, minCoefficients=1e-6)))
        # atoms planted against both edges: reflect-padding quirk of the local update (modeling.py:1046)
        D = norm(rs.standard_normal((8, 16)).astype(dt), axis=1)
        x = (0.05 * rs.standard_normal(200)).astype(np.float64)
        for c, p, k in zip([2.0, -1.5, 1.0, 3.0, -2.5], [2, 9, 100, 195, 199], [1, 5, 2, 7, 0]):
            s, e, es, ee = synth.centered_span(200, 16, p)
            x[s:e] += c * D[k][es:ee]
        cases.append(('%s_edges_T200_K8_W16_L12' % tag, x.astype(dt), D, dict(nbNonzeroCoefs=12)))
        # odd width at the edges
        D = norm(rs.standard_normal((6, 9)).astype(dt), axis=1)
        x = rs.standard_normal(40).astype(dt)
        cases.append(('%s_T40_K6_W9_L25' % tag, x, D, dict(nbNonzeroCoefs=25)))
    # a random family from its own stream (appended: the cases above keep their inputs): short signals with even and
    # odd widths whose pursuit keeps coming back to the borders (edge history, stale reflected sample of row T-1),
    # planted + noisy mid-size signals, every selection mode
    r2 = np.random.RandomState(20261004)
    for q in range(64):
        dt = np.float32 if q % 2 else np.float64
        tag = 'f32' if dt == np.float32 else 'f64'
        if q < 32:
            W = int(r2.choice([2, 4, 6, 8, 3, 5])); K = int(r2.randint(1, 10)); T = int(r2.randint(3 * W, 12 * W + 10)); F = 1
        else:
            W = int(r2.randint(4, 24)); K = int(r2.randint(2, 24)); T = int(r2.randint(4 * W, 300)); F = int(r2.choice([1, 1, 3]))
        D = r2.standard_normal((K, W) if F == 1 else (K, W, F)).astype(dt)
        D = norm(D, axis=tuple(range(1, D.ndim)))
        x = (0.05 * r2.standard_normal((T,) if F == 1 else (T, F))).astype(dt)
        D3 = D.reshape((K, W, -1)); x2 = x.reshape((T, -1))
        for _ in range(int(r2.randint(3, 10))):
            k = r2.randint(0, K); t0 = int(r2.choice([0, T - W, r2.randint(0, T - W + 1)]))
            x2[t0:t0 + W] += (r2.uniform(0.5, 2.0) * r2.choice([-1.0, 1.0]) * D3[k]).astype(dt)
        kw = {}
        mode = q % 4
        if mode == 1 and T >= 8:
            kw['nbBlocks'] = int(r2.randint(2, min(8, T // 3)))
        elif mode == 2:
            kw['nbBlocks'] = 'auto'
        if q % 3 == 0:
            kw['toleranceSnr'] = float(r2.uniform(8.0, 25.0)); kw['nbNonzeroCoefs'] = 40
        else:
            kw['nbNonzeroCoefs'] = int(r2.randint(5, 30))
        name = 'rand%02d_%s_T%d_K%d_W%d_F%d' % (q, tag, T, K, W, F)
        cases.append((name, x, D, kw))
    return cases


def pack_csc(prefix, m, out):
    m = m.tocoo()
    order = np.lexsort((m.row, m.col))
    out[prefix + '_row'] = m.row[order].astype(np.int32)
    out[prefix + '_col'] = m.col[order].astype(np.int32)
    out[prefix + '_data'] = m.data[order].astype(np.float64)


def gen_small():
    out = {}
    names = []
    for name, x, D, kw in small_cases():
        coefficients, residual, tr = run_reference_cmp(x, D, **kw)
        names.append(name)
        out[name + '__x'] = x
        out[name + '__D'] = D
        for key, val in kw.items():
            if key == 'weights':
                out[name + '__weights'] = val
            elif key == 'nbBlocks':
                out[name + '__nbBlocks'] = np.array(-1 if val == 'auto' else val)
            else:
                out[name + '__' + key] = np.array(val)
        out[name + '__t'] = tr['t']
        out[name + '__k'] = tr['k']
        out[name + '__c'] = np.asarray(tr['c'])
        out[name + '__residual'] = residual
        pack_csc(name + '__csc', coefficients, out)
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'cmp_small.npz'), **out)
    print('cmp_small.npz: %d cases' % len(names))


def gen_functions():
    ref = load_reference()
    m = ref.modeling
    norm = ref.utils.normalize
    rs = np.random.RandomState(777)
    out = {}
    names = []
    cmp = m.ConvolutionalMatchingPursuit()
    for dt in (np.float32, np.float64):
        tag = 'f32' if dt == np.float32 else 'f64'
        for (T, K, W, F) in [(50, 5, 4, 1), (50, 5, 7, 1), (40, 6, 8, 3), (33, 4, 5, 2), (200, 16, 32, 1)]:
            name = '%s_T%d_K%d_W%d_F%d' % (tag, T, K, W, F)
            names.append(name)
            D = norm(rs.standard_normal((K, W) if F == 1 else (K, W, F)).astype(dt), axis=tuple(range(1, 2 if F == 1 else 3)))
            x = rs.standard_normal((T,) if F == 1 else (T, F)).astype(dt)
            out[name + '__x'] = x
            out[name + '__D'] = D
            ip = m.convolve1d(x, D, padding='same')
            out[name + '__same'] = ip
            out[name + '__valid'] = m.convolve1d(x, D, padding='valid')
            # _selectBestAtoms on the reference's table, several modes (modeling.py:899-982)
            w = (0.25 + rs.random_sample(K)).astype(dt)
            modes = [(1, False, None), (1, False, w), (2, False, None), (2, True, None), (3, True, w),
                     (5, False, None), (5, True, None), ('auto', False, None), ('auto', True, w)]
            for i, (nb, off, ww) in enumerate(modes):
                atoms = cmp._selectBestAtoms(ip, W, nbBlocks=nb, offset=off, nullCoeffThres=1e-16, weights=ww)
                out['%s__sel%d_nb' % (name, i)] = np.array(-1 if nb == 'auto' else nb)
                out['%s__sel%d_offset' % (name, i)] = np.array(int(off))
                if ww is not None:
                    out['%s__sel%d_weights' % (name, i)] = ww
                out['%s__sel%d_t' % (name, i)] = np.array([a.position for a in atoms], dtype=np.int32)
                out['%s__sel%d_k' % (name, i)] = np.array([a.index for a in atoms], dtype=np.int32)
                out['%s__sel%d_c' % (name, i)] = np.array([a.coefficient for a in atoms], dtype=dt)
            out[name + '__nsel'] = np.array(len(modes))
            # _updateInnerProducts at both edges and in the interior (modeling.py:1018-1051):
            # perturb the residual on the atom's support (as a subtraction would), then let the
            # reference refresh the table
            positions = [0, 1, W // 2, T // 2, T - 2, T - 1]
            r = x.copy()
            ip2 = ip.copy()
            for j, p in enumerate(positions):
                atom = m.Atom(p, 0, 1.0, W)
                r2 = r.copy()
                s, e, es, ee = synth.centered_span(T, W, p)
                r2[s:e] += (0.5 * rs.standard_normal(r2[s:e].shape)).astype(dt)
                r = r2
                ip2 = cmp._updateInnerProducts(ip2, r.reshape((T, -1)), [atom], D.reshape((K, W, -1)))
                out['%s__upd%d_p' % (name, j)] = np.array(p)
                out['%s__upd%d_r' % (name, j)] = r.copy()
                out['%s__upd%d_ip' % (name, j)] = ip2.copy()
            out[name + '__nupd'] = np.array(len(positions))
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'functions.npz'), **out)
    print('functions.npz: %d cases' % len(names))


def gen_config():
    out = {}
    # config 1 (scripts/demo_csc.py-sized): T=4096, K=32, W=32, L0=64, f32, planted + noise
    D1 = synth.make_dictionary(32, 32, seed=1)
    out['config1__D_digest'] = np.array(synth.digest(D1))
    for kind in ('planted', 'noise'):
        x = synth.make_signal(D1, 4096, 0, kind=kind, nb_atoms=64, seed=1)
        t0 = time.time()
        coefficients, residual, tr = run_reference_cmp(x, D1, nbNonzeroCoefs=64)
        name = 'config1_%s' % kind
        out[name + '__x_digest'] = np.array(synth.digest(x))
        out[name + '__t'] = tr['t']; out[name + '__k'] = tr['k']; out[name + '__c'] = np.asarray(tr['c'])
        out[name + '__residual_energy'] = np.array(float(np.sum(np.square(residual.astype(np.float64)))))
        pack_csc(name + '__csc', coefficients, out)
        print(name, 'ref time %.2fs' % (time.time() - t0), 'n', len(tr['t']))
    # config 2: full size, 4 planted + 4 noise signals
    D2 = synth.make_dictionary(256, 64, seed=2)
    out['config2__D_digest'] = np.array(synth.digest(D2))
    for kind in ('planted', 'noise'):
        for idx in range(4):
            x = synth.make_signal(D2, 65536, idx, kind=kind, nb_atoms=256, seed=2)
            t0 = time.time()
            coefficients, residual, tr = run_reference_cmp(x, D2, nbNonzeroCoefs=256)
            name = 'config2_%s_%d' % (kind, idx)
            out[name + '__x_digest'] = np.array(synth.digest(x))
            out[name + '__t'] = tr['t']; out[name + '__k'] = tr['k']; out[name + '__c'] = np.asarray(tr['c'])
            out[name + '__residual_energy'] = np.array(float(np.sum(np.square(residual.astype(np.float64)))))
            pack_csc(name + '__csc', coefficients, out)
            print(name, 'ref time %.2fs' % (time.time() - t0), 'n', len(tr['t']))
    np.savez_compressed(os.path.join(OUT, 'cmp_config.npz'), **out)


def make_hsc_dictionaries(rs, dtype=np.float32):
    """Small 3-level raw dictionaries (no singletons): scales [16,31,61] => widths [16,16,31]."""
    scales = [16, 31, 61]
    D0 = rs.standard_normal((8, 16))
    D0 = (D0 / np.sqrt(np.sum(D0 ** 2, axis=1, keepdims=True))).astype(dtype)
    dicts = [D0]
    for (K, W, F) in [(6, 16, 8), (4, 31, 6)]:
        D = np.zeros((K, W, F))
        for k in range(K):
            for _ in range(3):
                D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
            D[k] /= np.sqrt(np.sum(D[k] ** 2))
        dicts.append(D.astype(dtype))
    return dicts, scales


def gen_hsc():
    """Hierarchical encoder (modeling.py:1427-1705, method='cmp') on a small 3-level dictionary."""
    ref = load_reference()
    rs = np.random.RandomState(4242)
    out = {}
    dicts, scales = make_hsc_dictionaries(rs)
    mld = ref.dataset.MultilevelDictionary.fromRawDictionaries(dicts, scales)
    mlds = mld.withSingletonBases()
    for l, D in enumerate(dicts):
        out['raw%d' % l] = D
    out['scales'] = np.array(scales)
    for l in range(3):
        out['single_raw%d' % l] = mlds.getRawDictionary(l)
        out['single_rep%d' % l] = mlds.getMultiscaleDictionaries()[l]
        out['rep%d' % l] = mld.getMultiscaleDictionaries()[l]
    out['counts'] = np.array(mlds.counts)
    out['countsNoSingletons'] = np.array(mlds.countsNoSingletons)
    # signal: events of all levels through the input-level representations, plus a little noise
    T = 512
    x = 0.01 * rs.standard_normal(T)
    for l in range(3):
        rep = mld.getMultiscaleDictionaries()[l]
        for _ in range(6):
            i = rs.randint(0, rep.shape[0]); t = rs.randint(40, T - 40); c = rs.uniform(0.5, 2.0) * rs.choice([-1.0, 1.0])
            s, e, es, ee = synth.centered_span(T, rep.shape[1], t)
            x[s:e] += c * rep[i][es:ee]
    x = x.astype(np.float32)
    out['x'] = x
    cases = [('a', dict(toleranceSnr=15.0, nbBlocks=1, singletonWeight=0.9)),
             ('b', dict(toleranceSnr=[20.0, 25.0, 30.0], nbBlocks=4, singletonWeight=0.5, returnDistributed=False)),
             ('c', dict(toleranceSnr=[10.0, 40.0, 40.0], nbBlocks='auto', singletonWeight=0.95)),
             ('d', dict(toleranceSnr=[10.0, 20.0, 20.0], nbBlocks=4, singletonWeight=0.5))]      # d: method='locomp'
    names = []
    for name, kw in cases:
        hcmp = ref.modeling.HierarchicalConvolutionalMatchingPursuit(method='locomp' if name == 'd' else 'cmp')
        hcsc = ref.modeling.HierarchicalConvolutionalSparseCoder(mld, hcmp)
        coefficients, residual = hcsc.encode(x, **kw)
        names.append(name)
        snr = kw['toleranceSnr']
        out['case_%s__toleranceSnr' % name] = np.atleast_1d(np.array(snr, dtype=np.float64))
        out['case_%s__snr_is_list' % name] = np.array(int(isinstance(snr, list)))
        out['case_%s__nbBlocks' % name] = np.array(-1 if kw['nbBlocks'] == 'auto' else kw['nbBlocks'])
        out['case_%s__singletonWeight' % name] = np.array(kw['singletonWeight'])
        out['case_%s__returnDistributed' % name] = np.array(int(kw.get('returnDistributed', True)))
        out['case_%s__residual' % name] = residual
        for l, c in enumerate(coefficients):
            pack_csc('case_%s__level%d' % (name, l), scipy_sparse(c), out)
        out['case_%s__recon' % name] = hcsc.reconstruct(coefficients)
        # encodeFromLevel: restart from the level-0 forward result
        raw = hcmp._forwardPhase(x, hcsc.multilevelDict, kw['toleranceSnr'], kw['nbBlocks'], kw['singletonWeight'])
        cont = hcsc.encodeFromLevel(x, raw[:1], toleranceSnr=kw['toleranceSnr'], nbBlocks=kw['nbBlocks'],
                                    singletonWeight=kw['singletonWeight'], returnDistributed=kw.get('returnDistributed', True))
        for l, c in enumerate(cont):
            pack_csc('case_%s__fromlevel%d' % (name, l), scipy_sparse(c), out)
        if name == 'a':
            # events wire format (dataset.py:798-824) of the distributed coefficients
            ev = ref.dataset.convertSparseMatricesToEvents(coefficients)
            out['case_a__events_t'] = ev['f0']; out['case_a__events_l'] = ev['f1']
            out['case_a__events_i'] = ev['f2']; out['case_a__events_c'] = ev['f3']
            back = ref.dataset.convertEventsToSparseMatrices(ev, [c.shape[1] for c in coefficients], coefficients[0].shape[0])
            for l, bm in enumerate(back):
                pack_csc('case_a__back%d' % l, bm, out)
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'hsc_small.npz'), **out)
    print('hsc_small.npz: %d cases' % len(names))


def gen_hsc_medium():
    """Hierarchical encoder of the REAL reference on generated data at a realistic level-1 shape: Perlin
    dictionary (scales [32, 64], 24 + 20 patterns), Poisson-event signal of 8192 samples, blocked selection.
    The dictionary and the signal are regenerated by the tests from the seeds (tests/test_dataset_synthesis.py
    pins the generators), only digests and the reference's coefficients are stored."""
    import hashlib
    ref = load_reference()
    out = {}
    np.random.seed(77)
    mld = ref.dataset.MultilevelDictionaryGenerator().generate(scales=[32, 64], counts=[24, 20], decompositionSize=3,
                                                               multilevelDecomposition=False, maxNbPatternsConsecutiveRejected=50)
    np.random.seed(78)
    gen = ref.dataset.SignalGenerator(mld, [0.004, 0.004])
    events = gen.generateEvents(8192)
    x = gen.generateSignalFromEvents(events, nbSamples=8192)
    out['x_sha256'] = np.array(hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest())
    out['nevents'] = np.int64(len(events))
    for name, kw in (('a', dict(toleranceSnr=[20.0, 25.0], nbBlocks=8, singletonWeight=0.9)),
                     ('b', dict(toleranceSnr=[25.0, 30.0], nbBlocks='auto', singletonWeight=0.95, returnDistributed=False))):
        t0 = time.time()
        hcsc = ref.modeling.HierarchicalConvolutionalSparseCoder(mld, ref.modeling.HierarchicalConvolutionalMatchingPursuit(method='cmp'))
        coefficients, residual = hcsc.encode(x, **kw)
        for l, c in enumerate(coefficients):
            pack_csc('case_%s__level%d' % (name, l), scipy_sparse(c), out)
        out['case_%s__residual_energy' % name] = np.float64(np.sum(np.square(np.asarray(residual, dtype=np.float64))))
        print('hsc_medium', name, [c.nnz for c in coefficients], '%.1f s' % (time.time() - t0))
    np.savez_compressed(os.path.join(OUT, 'hsc_medium.npz'), **out)
    print('wrote hsc_medium.npz', os.path.getsize(os.path.join(OUT, 'hsc_medium.npz')), 'bytes')


def gen_hsc_config4():
    """Hierarchical encoder of the REAL reference at the dictionary dimensions of BASELINE config 4 (level 0: 256 x 64,
    level 1: (256 singletons + 128) x W1 x 256; hsc_amd.synth.make_hierarchy), one 8192-sample signal per case: W1=16 is the
    exact shape, W1=17 the nearest one whose hierarchy reconstructs (see make_hierarchy).  Inputs are regenerated by the
    tests from the seeds; digests and the reference's coefficients are stored.  Also: reconstructSignal's DENSE branch
    (modeling.py:247-258, fftconvolve) on a small seeded case, which the engine answers by overlap-add."""
    import copy
    ref = load_reference()
    out = {}
    kw = dict(toleranceSnr=[30.0, 40.0], nbBlocks=10, singletonWeight=0.95)
    for W1 in (16, 17):
        D0, decompositions, scales = synth.make_hierarchy_parts(W1=W1, seed=4)
        mld = ref.dataset.MultilevelDictionary.fromDecompositions(D0, copy.deepcopy(decompositions), np.array(scales))
        own = synth.make_hierarchy(W1=W1, seed=4)
        x = synth.make_hierarchy_signal(own, 8192, 0, seed=4)
        out['w%d__x_digest' % W1] = np.array(synth.digest(x))
        out['w%d__dict_digest' % W1] = np.array(synth.digest(*own.withSingletonBases().dictionaries))
        mlds = mld.withSingletonBases()
        assert all(np.array_equal(a, b) for a, b in zip(mlds.dictionaries, own.withSingletonBases().dictionaries))
        t0 = time.time()
        hcsc = ref.modeling.HierarchicalConvolutionalSparseCoder(mlds, ref.modeling.HierarchicalConvolutionalMatchingPursuit(method='cmp'))
        coefficients, residual = hcsc.encode(x, **kw)
        for l, c in enumerate(coefficients):
            pack_csc('w%d__level%d' % (W1, l), scipy_sparse(c), out)
        e = float(np.sum(np.square(np.asarray(residual, dtype=np.float64))))
        out['w%d__residual_energy' % W1] = np.float64(e)
        print('hsc_config4 W1=%d' % W1, [c.nnz for c in coefficients], 'SNR %.2f dB' % (10 * np.log10(np.sum(x.astype(np.float64) ** 2) / e)),
              '%.1f s' % (time.time() - t0))
    # dense branch of reconstructSignal
    rs = np.random.RandomState(20261004)
    for name, (T, K, W, F) in (('dense1d', (96, 5, 8, 1)), ('dense1d_odd', (80, 4, 9, 1)), ('dense2d', (64, 3, 6, 4))):
        D = rs.standard_normal((K, W) if F == 1 else (K, W, F))
        C = np.zeros((T, K))
        for _ in range(12):
            C[rs.randint(0, T), rs.randint(0, K)] = rs.uniform(-2, 2)
        out['%s__D' % name] = D; out['%s__C' % name] = C
        out['%s__signal' % name] = np.asarray(ref.modeling.reconstructSignal(C, D))
        import scipy.sparse
        out['%s__signal_sparse' % name] = np.asarray(ref.modeling.reconstructSignal(scipy.sparse.csc_matrix(C), D))
    np.savez_compressed(os.path.join(OUT, 'hsc_config4.npz'), **out)
    print('wrote hsc_config4.npz', os.path.getsize(os.path.join(OUT, 'hsc_config4.npz')), 'bytes')


def gen_mld_pickle():
    """A multilevel dictionary (with singleton bases) saved by the REAL reference's MultilevelDictionary.save
    (dataset.py:389-396; cPickle protocol 2 under Python 2): tests/golden/mld_reference.pkl.  Data only: the class is
    pickled by reference (module path `hsc.dataset`), the arrays by value."""
    import pickle
    import types
    ref = load_reference()
    z = np.load(os.path.join(OUT, 'hsc_small.npz'))
    mld = ref.dataset.MultilevelDictionary.fromRawDictionaries([z['raw0'], z['raw1'], z['raw2']], [int(v) for v in z['scales']])
    mlds = mld.withSingletonBases()
    saved = {k: sys.modules.get(k) for k in ('hsc', 'hsc.dataset')}
    sys.modules.setdefault('hsc', types.ModuleType('hsc'))
    sys.modules['hsc.dataset'] = ref.dataset              # the in-memory module the loader built: lets pickle resolve the class path
    try:
        with open(os.path.join(OUT, 'mld_reference.pkl'), 'wb') as f:
            pickle.dump(mlds, f, protocol=2)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    print('wrote mld_reference.pkl', os.path.getsize(os.path.join(OUT, 'mld_reference.pkl')), 'bytes')


def gen_locomp():
    """LoCOMP (modeling.py:1191-1425) on small seeded problems."""
    ref = load_reference()
    norm = ref.utils.normalize
    rs = np.random.RandomState(1357)
    out = {}
    names = []

    def planted(D, T, events):
        x = np.zeros((T,) + D.shape[2:], dtype=np.float64)
        for c, p, k in events:
            s, e, es, ee = synth.centered_span(T, D.shape[1], p)
            x[s:e] += c * D[k][es:ee]
        return x

    cases = []
    ev = list(zip([1.0, 1.0, 0.5, 1.0, 0.75, 2.0], [32, 48, 64, 96, 128, 192], [0, 3, 1, 0, 2, 2]))
    for dt, tag in ((np.float64, 'f64'), (np.float32, 'f32')):
        D = norm(rs.random_sample((4, 32)).astype(dt), axis=1)
        cases.append(('%s_planted_1d' % tag, planted(D, 256, ev).astype(dt), D, dict(minCoefficients=1e-10)))
        D3 = norm(rs.random_sample((4, 32, 7)).astype(dt), axis=(1, 2))
        cases.append(('%s_planted_2d' % tag, planted(D3, 256, ev).astype(dt), D3, dict(minCoefficients=1e-10)))
        D = norm(rs.standard_normal((16, 15)).astype(dt), axis=1)
        x = rs.standard_normal(256).astype(dt)
        for nb in (1, 2, 8, 'auto'):
            cases.append(('%s_T256_K16_W15_snr5_nb%s' % (tag, nb), x, D, dict(toleranceSnr=5.0, nbBlocks=nb)))
        cases.append(('%s_T256_K16_W15_L12' % tag, x, D, dict(nbNonzeroCoefs=12)))
        w = np.ones(16, dtype=dt); w[:4] = 0.5
        cases.append(('%s_T256_K16_W15_snr8_weights' % tag, x, D, dict(toleranceSnr=8.0, nbBlocks=4, weights=w)))
    # a random family: odd / even widths, multi-feature, both dtypes, every selection mode
    for q in range(28):
        dt = np.float64 if q % 2 == 0 else np.float32
        T = int(rs.randint(100, 500)); K = int(rs.randint(2, 20)); W = int(rs.randint(3, 24)); F = int(rs.choice([1, 1, 3]))
        D = rs.standard_normal((K, W) if F == 1 else (K, W, F)).astype(dt)
        D = norm(D, axis=tuple(range(1, D.ndim)))
        x = (0.05 * rs.standard_normal((T,) if F == 1 else (T, F))).astype(dt)
        D3 = D.reshape((K, W, -1)); x2 = x.reshape((T, -1))
        for _ in range(int(rs.randint(3, 12))):
            k = rs.randint(0, K); t0 = rs.randint(0, T - W)
            x2[t0:t0 + W] += (rs.uniform(0.5, 2.0) * rs.choice([-1.0, 1.0]) * D3[k]).astype(dt)
        kw = dict(nbBlocks=[1, 2, int(rs.randint(3, 9)), 'auto'][q % 4])
        if q % 3 == 0:
            kw['nbNonzeroCoefs'] = int(rs.randint(3, 15))
        else:
            kw['toleranceSnr'] = float(rs.uniform(5.0, 18.0))
        if q % 5 == 0:
            w = np.ones(K, dtype=dt); w[:K // 2] = 0.7
            kw['weights'] = w
        cases.append(('rand%02d_%s_T%d_K%d_W%d_F%d' % (q, 'f64' if dt == np.float64 else 'f32', T, K, W, F), x, D, kw))
    for name, x, D, kw in cases:
        coefficients, residual = ref.modeling.LoCOMP().computeCoefficients(x, D, **kw)
        names.append(name)
        out[name + '__x'] = x
        out[name + '__D'] = D
        for key, val in kw.items():
            if key == 'weights':
                out[name + '__weights'] = val
            elif key == 'nbBlocks':
                out[name + '__nbBlocks'] = np.array(-1 if val == 'auto' else val)
            else:
                out[name + '__' + key] = np.array(val)
        out[name + '__residual'] = residual
        pack_csc(name + '__csc', coefficients, out)
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'locomp_small.npz'), **out)
    print('locomp_small.npz: %d cases' % len(names))



class _PinvProbe(object):
    """np.linalg.pinv of the reference's LoCOMP (modeling.py:1326), observed: per call the group size, the singular values the
    pseudo-inverse keeps (rcond = 1e-15 of the largest, in the matrix's dtype) and how many it cuts."""

    def __init__(self, ref):
        self.ref, self.rec = ref, []
        self.orig = np.linalg.pinv

    def __enter__(self):
        def pinv(a, *args, **kw):
            s = np.linalg.svd(a, compute_uv=False)            # (the dtype the reference's own call works in)
            keep = s > 1e-15 * s[0]
            self.rec.append((a.shape[0], a.shape[1], int(np.sum(~keep)) + max(0, a.shape[0] - len(s)), float(s[keep][-1] / s[0])))
            return self.orig(a, *args, **kw)
        self.ref.modeling.np.linalg.pinv = pinv
        del self.rec[:]
        return self

    def __exit__(self, *exc):
        self.ref.modeling.np.linalg.pinv = self.orig

    def summary(self, out, prefix):
        r = np.array(self.rec, dtype=np.float64) if self.rec else np.zeros((0, 4))
        out[prefix + '__groups'] = np.int64(len(r))
        out[prefix + '__group_max'] = np.int64(r[:, 0].max() if len(r) else 0)
        out[prefix + '__groups_cut'] = np.int64(int((r[:, 2] > 0).sum()) if len(r) else 0)       # groups with a cut singular value
        out[prefix + '__min_rel_sigma_kept'] = np.float64(r[:, 3].min() if len(r) else 1.0)         # 1 / condition of the worst group
        return r


def rank_deficient_case(seed, dt, F=6, W=7, ncomp=5, T=200, nev=20):
    """A multi-feature dictionary of singletons (one cell) and composites (two cells), and a signal in which every planted
    composite comes with the singletons of its two cells: the greedy selection puts all three into one group, whose local
    dictionary is exactly rank deficient (hierarchical levels >= 1 look like this: hsc/dataset.py:826-860)."""
    rs = np.random.RandomState(seed)
    K = F + ncomp
    D = np.zeros((K, W, F))
    c = (W - 1) // 2
    for f in range(F):
        D[f, c, f] = 1.0
    for k in range(F, K):
        for _ in range(2):
            D[k, rs.randint(0, W), rs.randint(0, F)] = rs.uniform(0.5, 1.5) * rs.choice([-1.0, 1.0])
        D[k] /= np.sqrt(np.sum(D[k] ** 2))
    x = np.zeros((T, F))
    for _ in range(nev):
        k = rs.randint(0, K); t = rs.randint(0, T - W)
        x[t:t + W] += rs.uniform(0.5, 2.0) * rs.choice([-1.0, 1.0]) * D[k]
        if k >= F:
            for (w_, f_) in zip(*np.nonzero(D[k])):
                x[t + w_, f_] += rs.uniform(0.5, 2.0) * rs.choice([-1.0, 1.0])
    x += 0.01 * rs.standard_normal((T, F)) * (rs.rand(T, F) < 0.1)
    w = np.ones(K); w[:F] = 0.9
    return x.astype(dt), D.astype(dt), dict(toleranceSnr=40.0, nbBlocks=2, weights=w.astype(dt), nbNonzeroCoefs=150)


def gen_locomp_hier():
    """The reference's DEFAULT hierarchical method (modeling.py:1429 method='locomp') and its re-fit on rank-deficient groups:
    tests/golden/locomp_hier.npz.
      c4w16 / c4w17   2-level encode at the BASELINE config-4 dictionary dimensions, 8192 samples (inputs by seed: hsc_amd.synth)
      gen3 / gen3_f64 3-level encode on a generated dictionary (Perlin, scales [16, 32, 64]; inputs by numpy seed), float32 as
                      generated and float64 (the reference's default dtype)
      rd_*            single-level LoCOMP on groups that are exactly rank deficient (a composite atom beside its singletons):
                      np.linalg.pinv cuts the vanishing singular value and returns the minimum-norm re-fit
      soak_*          the draws of tests/test_gpu_fuzz.py::_draw whose groups fill their whole (tiny) signal
    Every case carries what the reference's pseudo-inverse saw: number of groups, largest group, groups with a cut singular value,
    the smallest kept singular value relative to the largest (= 1 / condition of the worst group)."""
    import copy
    import hashlib
    import logging
    logging.disable(logging.WARNING)
    ref = load_reference()
    out = {}
    names = []

    def pack_levels(name, coefficients, residual, x):
        for l, c in enumerate(coefficients):
            pack_csc('%s__level%d' % (name, l), scipy_sparse(c), out)
        e = float(np.sum(np.square(np.asarray(residual, dtype=np.float64))))
        out[name + '__residual_energy'] = np.float64(e)
        out[name + '__nlevels'] = np.int64(len(coefficients))
        return 10 * np.log10(np.sum(np.asarray(x, dtype=np.float64) ** 2) / e)

    kw4 = dict(toleranceSnr=[30.0, 40.0], nbBlocks=10, singletonWeight=0.95)
    for W1 in (16, 17):
        name = 'c4w%d' % W1
        D0, decompositions, scales = synth.make_hierarchy_parts(W1=W1, seed=4)
        mld = ref.dataset.MultilevelDictionary.fromDecompositions(D0, copy.deepcopy(decompositions), np.array(scales))
        own = synth.make_hierarchy(W1=W1, seed=4)
        x = synth.make_hierarchy_signal(own, 8192, 0, seed=4)
        out[name + '__x_digest'] = np.array(synth.digest(x))
        mlds = mld.withSingletonBases()
        t0 = time.time()
        with _PinvProbe(ref) as probe:
            hcsc = ref.modeling.HierarchicalConvolutionalSparseCoder(mlds, ref.modeling.HierarchicalConvolutionalMatchingPursuit(method='locomp'))
            coefficients, residual = hcsc.encode(x, **kw4)
        probe.summary(out, name)
        snr = pack_levels(name, coefficients, residual, x)
        names.append(name)
        print(name, [c.nnz for c in coefficients], 'SNR %.2f dB' % snr, 'groups', int(out[name + '__groups']), 'max', int(out[name + '__group_max']),
              'cut', int(out[name + '__groups_cut']), 'min rel sigma %.3g' % float(out[name + '__min_rel_sigma_kept']), '%.1f s' % (time.time() - t0))

    kw3 = dict(toleranceSnr=[20.0, 30.0, 30.0], nbBlocks=10, singletonWeight=0.9)
    for name, dt in (('gen3', np.float32), ('gen3_f64', np.float64)):
        np.random.seed(77)
        mld = ref.dataset.MultilevelDictionaryGenerator().generate(scales=[16, 32, 64], counts=[16, 24, 32], decompositionSize=3,
                                                                   multilevelDecomposition=False, maxNbPatternsConsecutiveRejected=50)
        np.random.seed(78)
        gen = ref.dataset.SignalGenerator(mld, [0.004, 0.004, 0.004])
        events = gen.generateEvents(4096)
        x = gen.generateSignalFromEvents(events, nbSamples=4096)
        if dt == np.float64:
            dec64 = [[[d[0], d[1], d[2], np.asarray(d[3], dtype=np.float64)] for d in lev] for lev in copy.deepcopy(mld.decompositions)]
            mld = ref.dataset.MultilevelDictionary.fromDecompositions(mld.dictionaries[0].astype(np.float64), dec64, mld.scales)
            assert all(d.dtype == np.float64 for d in mld.dictionaries)
            x = x.astype(np.float64)
        out[name + '__x_sha256'] = np.array(hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest())
        out[name + '__nevents'] = np.int64(len(events))
        t0 = time.time()
        with _PinvProbe(ref) as probe:
            hcsc = ref.modeling.HierarchicalConvolutionalSparseCoder(mld, ref.modeling.HierarchicalConvolutionalMatchingPursuit(method='locomp'))
            coefficients, residual = hcsc.encode(x, **kw3)
        probe.summary(out, name)
        snr = pack_levels(name, coefficients, residual, x)
        names.append(name)
        print(name, x.dtype, [c.nnz for c in coefficients], 'SNR %.2f dB' % snr, 'groups', int(out[name + '__groups']), 'max', int(out[name + '__group_max']),
              'cut', int(out[name + '__groups_cut']), 'min rel sigma %.3g' % float(out[name + '__min_rel_sigma_kept']), '%.1f s' % (time.time() - t0))

    # single-level cases
    singles = []
    for dt, tag in ((np.float64, 'f64'), (np.float32, 'f32')):
        found = 0
        for seed in range(400):
            if found >= 5:
                break
            x, D, kw = rank_deficient_case(seed, dt)
            with _PinvProbe(ref) as probe:
                coef, res = ref.modeling.LoCOMP().computeCoefficients(x, D, **kw)
            r = np.array(probe.rec, dtype=np.float64) if probe.rec else np.zeros((0, 4))
            # keep the draws with at least one cut singular value whose KEPT part is well conditioned (the comparison is then at
            # round-off level, and the cut itself unambiguous: the vanishing value is exactly 0 or below 1e-15 by orders)
            if len(r) and (r[:, 2] > 0).any() and r[:, 3].min() > 1e-3 and np.all(np.isfinite(res)):
                singles.append(('rd_%s_s%d' % (tag, seed), x, D, kw, coef, res, list(probe.rec)))
                found += 1
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_gpu_fuzz as fz
    for i in (29, 53, 117):
        x, D, kw = fz._draw(i)
        kw = dict(kw)
        if kw.get('nbNonzeroCoefs', 0) > 40:
            kw['nbNonzeroCoefs'] = 40
        with _PinvProbe(ref) as probe:
            coef, res = ref.modeling.LoCOMP().computeCoefficients(x, D, **kw)
        singles.append(('soak_%d' % i, x, D, kw, coef, res, list(probe.rec)))
    for name, x, D, kw, coef, res, rec in singles:
        names.append(name)
        out[name + '__x'] = x; out[name + '__D'] = D
        for key, val in kw.items():
            if key == 'weights':
                out[name + '__weights'] = val
            elif key == 'nbBlocks':
                out[name + '__nbBlocks'] = np.array(-1 if val == 'auto' else val)
            elif val is None:
                out[name + '__%s_is_none' % key] = np.array(1)
            else:
                out[name + '__' + key] = np.array(val)
        out[name + '__residual'] = res
        pack_csc(name + '__csc', coef, out)
        r = np.array(rec, dtype=np.float64) if rec else np.zeros((0, 4))
        out[name + '__groups'] = np.int64(len(r)); out[name + '__group_max'] = np.int64(r[:, 0].max() if len(r) else 0)
        out[name + '__groups_cut'] = np.int64(int((r[:, 2] > 0).sum()) if len(r) else 0)
        out[name + '__min_rel_sigma_kept'] = np.float64(r[:, 3].min() if len(r) else 1.0)
        print(name, x.shape, D.shape, x.dtype, 'nnz', coef.nnz, 'groups', len(r), 'cut', int(out[name + '__groups_cut']),
              'min rel sigma kept %.3g' % float(out[name + '__min_rel_sigma_kept']))
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'locomp_hier.npz'), **out)
    print('wrote locomp_hier.npz', os.path.getsize(os.path.join(OUT, 'locomp_hier.npz')), 'bytes')
    logging.disable(logging.NOTSET)


sys.path.insert(0, os.path.join(ROOT, 'tests'))
from golden_util import SYNTH_CASES, LEARN_CASES, learn_signal  # noqa: E402  (case tables shared with the tests)


def gen_synth():
    """Dataset synthesis of the REAL reference under fixed seeds (the generators draw from numpy's global
    RandomState)."""
    ref = load_reference()
    out = {'names': np.array(sorted(SYNTH_CASES))}
    for name, (kw, dseed, rate, eseed, n) in sorted(SYNTH_CASES.items()):
        np.random.seed(dseed)
        mld = ref.dataset.MultilevelDictionaryGenerator().generate(**kw)
        out[name + '__nlevels'] = np.int64(mld.getNbLevels())
        out[name + '__singletons'] = np.int64(mld.hasSingletonBases)
        for l in range(mld.getNbLevels()):
            out['%s__raw%d' % (name, l)] = mld.dictionaries[l]
            out['%s__rep%d' % (name, l)] = mld.representations[l]
            if l > 0:
                for j, entry in enumerate(mld.decompositions[l - 1]):
                    for q, part in enumerate(entry):
                        out['%s__dec%d_%d_%d' % (name, l, j, q)] = np.asarray(part)
        rates = [rate] * mld.getNbLevels()
        for tag, ratio in (('plain', None), ('scaled', 0.25)):
            np.random.seed(eseed)
            gen = ref.dataset.SignalGenerator(mld, rates)
            res = gen.generateEvents(n, ratio)
            events = res if ratio is None else res[0]
            if ratio is not None:
                out['%s__%s_rates' % (name, tag)] = np.asarray(res[1], dtype=np.float64)
            for f in events.dtype.names:
                out['%s__%s_ev_%s' % (name, tag, f)] = events[f]
            out['%s__%s_signal' % (name, tag)] = gen.generateSignalFromEvents(events, nbSamples=n)
            out['%s__%s_autolen' % (name, tag)] = np.int64(len(gen.generateSignalFromEvents(events)))
        print('synth', name, [d.shape for d in mld.dictionaries], len(events), 'events')
    path = os.path.join(OUT, 'synth_small.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


def gen_learn():
    """Convolutional k-means learner of the REAL reference (modeling.py:420-524) under fixed seeds, plus the
    window-assignment step on its own (convolve1d_batch + arg-max, :454-460)."""
    ref = load_reference()
    out = {'names': np.array(sorted(LEARN_CASES))}
    for name, (sig, k, w, seed, kw) in sorted(LEARN_CASES.items()):
        data = learn_signal(sig)
        np.random.seed(seed)
        out[name + '__D'] = ref.modeling.ConvolutionalDictionaryLearner(k, w, algorithm='kmean').train(data, **kw)
        # one assignment step on fixed windows / dictionary
        rs = np.random.RandomState(seed + 100)
        idx = rs.randint(0, data.shape[0] - 2 * w, size=64)
        windows = ref.modeling.extractWindows(data, idx, 2 * w)
        D0 = out[name + '__D']
        ip = ref.modeling.convolve1d_batch(windows, D0, padding='valid')
        flat = np.argmax(np.abs(ip.reshape(ip.shape[0], -1)), axis=1)
        t, kk = np.unravel_index(flat, (ip.shape[1], ip.shape[2]))
        out[name + '__win_idx'] = idx
        out[name + '__assign_t'] = t.astype(np.int64); out[name + '__assign_k'] = kk.astype(np.int64)
        out[name + '__assign_c'] = ip[np.arange(len(idx)), t, kk]
        print('learn', name, out[name + '__D'].shape, out[name + '__D'].dtype)
    # K-SVD on top of the matching pursuit (modeling.py:526-641); atoms are defined up to sign (SVD)
    np.random.seed(3)
    out['ksvd_1d__D'] = ref.modeling.ConvolutionalDictionaryLearner(6, 16, algorithm='ksvd').train(
        learn_signal('planted_1d'), method='cmp', maxIterations=2, nbNonzeroCoefs=100, toleranceSnr=None)
    np.random.seed(7)
    out['samples_2d__D'] = ref.modeling.ConvolutionalDictionaryLearner(5, 9, algorithm='samples').train(learn_signal('sparse_2d'), avoidSingletons=True)
    path = os.path.join(OUT, 'learn_small.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


def scipy_sparse(c):
    import scipy.sparse
    return c if scipy.sparse.issparse(c) else scipy.sparse.csc_matrix(c)


if __name__ == '__main__':
    assert load_reference() is not None, 'the reference is not available in this environment'
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['small', 'functions', 'config', 'hsc', 'hscmed', 'hsc4', 'mldpkl', 'locomp', 'locomphier', 'synth', 'learn']
    if 'small' in which:
        gen_small()
    if 'functions' in which:
        gen_functions()
    if 'config' in which:
        gen_config()
    if 'hsc' in which:
        gen_hsc()
    if 'hscmed' in which:
        gen_hsc_medium()
    if 'hsc4' in which:
        gen_hsc_config4()
    if 'mldpkl' in which:
        gen_mld_pickle()
    if 'locomp' in which:
        gen_locomp()
    if 'locomphier' in which:
        gen_locomp_hier()
    if 'synth' in which:
        gen_synth()
    if 'learn' in which:
        gen_learn()
