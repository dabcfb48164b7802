"""Own counterpart of the reference's dataset scripts (scripts/generate_dataset.py:40-100,
BASELINE config 5): a multilevel dictionary of Perlin base atoms + random decompositions and
Poisson-event signals rendered from it, under fixed seeds so the GPU box regenerates the same data.

    python tools/generate_dataset.py --scales 32 64 128 --overcomplete 4 --samples 65536 --signals 8 --out /tmp/ds

Writes <out>/multilevel-dict.pkl and <out>/dataset.npz (signals [B,T] float32, events per signal, rates).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsc_amd.dataset import MultilevelDictionaryGenerator, SignalGenerator, scalesToWindowSizes  # noqa: E402


def build(scales, overcomplete=4, decompositionSize=3, seed=5, patience=1000, counts=None):
    """K_l = overcomplete x (window width of level l) patterns per level unless counts is given."""
    widths = scalesToWindowSizes(scales)
    counts = [int(overcomplete * w) for w in widths] if counts is None else counts
    rs = np.random.RandomState(0x48534300 + seed)
    gen = MultilevelDictionaryGenerator(rs)
    return gen.generate(scales, counts, decompositionSize=decompositionSize, positionSampling='random', weightSampling='random',
                        multilevelDecomposition=False, maxNbPatternsConsecutiveRejected=patience, nonNegativity=False)


def signals(mld, nbSignals, nbSamples, rate=5e-4, compression=0.25, seed=5, first=0):
    """One event stream + rendered signal per index first .. first+nbSignals-1, each from its own RandomState stream
    (shard-invariant).  The common rate scaling for the target compression ratio (hsc/dataset.py:686-706) is estimated
    once: it depends on the dictionary and the length only."""
    xs, evs = [], []
    rates = rate * np.ones(mld.getNbLevels())
    if compression is not None:
        rates = SignalGenerator(mld, rates)._estimateOptimalRates(compression, nbSamples)
    for b in range(first, first + nbSignals):
        rs = np.random.RandomState((0x48534300 + seed) * 1000003 % (2 ** 31) + b)
        gen = SignalGenerator(mld, rates, rng=rs)
        events = gen.generateEvents(nbSamples)
        xs.append(gen.generateSignalFromEvents(events, nbSamples=nbSamples))
        evs.append(events)
    return np.stack(xs), evs, rates


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--scales', type=int, nargs='+', default=[32, 64, 128])
    ap.add_argument('--overcomplete', type=float, default=4.0)
    ap.add_argument('--samples', type=int, default=65536)
    ap.add_argument('--signals', type=int, default=8)
    ap.add_argument('--seed', type=int, default=5)
    ap.add_argument('--out', default='.')
    a = ap.parse_args()
    t0 = time.time()
    mld = build(a.scales, a.overcomplete, seed=a.seed)
    print('dictionary: counts %s, raw shapes %s (%.1f s)' % (mld.counts.tolist(), [d.shape for d in mld.dictionaries], time.time() - t0))
    xs, evs, rates = signals(mld, a.signals, a.samples, seed=a.seed)
    os.makedirs(a.out, exist_ok=True)
    mld.save(os.path.join(a.out, 'multilevel-dict.pkl'))
    np.savez_compressed(os.path.join(a.out, 'dataset.npz'), signals=xs, rates=rates,
                        **{'events%d' % b: e for b, e in enumerate(evs)})
    print('signals %s, %d events per signal on average, rates %s (%.1f s)' % (xs.shape, np.mean([len(e) for e in evs]), rates, time.time() - t0))
