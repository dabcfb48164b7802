"""Own counterpart of scripts/learn_mlcsc_dataset.py:84-116: per level, learn a dictionary by convolutional
k-means on the current representation (window assignment on the GPU), encode the training signal with the
hierarchical matching pursuit built so far, and hand the last level's coefficients to the next level.
Prints the time of each stage.  Data: tools/generate_dataset.py (Perlin dictionary, Poisson events)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generate_dataset as gd
from hsc_amd.dataset import MultilevelDictionary, addSingletonBases, scalesToWindowSizes
from hsc_amd.learning import ConvolutionalDictionaryLearner
from hsc_amd.modeling import HierarchicalConvolutionalMatchingPursuit, HierarchicalConvolutionalSparseCoder

T = int(os.environ.get('T', '20000'))
truth = gd.build([32, 64, 128], counts=[16, 32, 64], patience=200)   # (scale 96 leaves no room to spread scale-64 parts)
xs, _, _ = gd.signals(truth, 1, T, rate=2e-3, compression=None)
train = xs[0]
counts, scales, snr = np.array([16, 32, 64]), np.array([32, 64, 96]), 10.0
widths = scalesToWindowSizes(scales)
np.random.seed(1)
dictionaries, inp, coefficients = [], train, None
for level, (k, w) in enumerate(zip(counts, widths)):
    nfeat = 1 if inp.ndim == 1 else inp.shape[1]
    t0 = time.perf_counter()
    D = ConvolutionalDictionaryLearner(k, w, algorithm='kmean').train(inp, nbRandomWindows=10000, maxIterations=10, tolerance=0.0,
                                                                      resetMethod='random_samples')
    t1 = time.perf_counter()
    dictionaries.append(D)
    if level > 0:
        mld = MultilevelDictionary.fromRawDictionaries(addSingletonBases(dictionaries), scales[:level + 1], hasSingletonBases=True)
    else:
        mld = MultilevelDictionary.fromRawDictionaries(dictionaries, scales[:1])
    hcsc = HierarchicalConvolutionalSparseCoder(mld, HierarchicalConvolutionalMatchingPursuit(method='cmp'))
    if level == 0:
        coefficients, residual = hcsc.encode(train, toleranceSnr=snr, nbBlocks=10, singletonWeight=0.95, returnDistributed=False)
    elif level < len(counts) - 1:
        coefficients = hcsc.encodeFromLevel(train, coefficients, toleranceSnr=snr, nbBlocks=10, singletonWeight=0.95, returnDistributed=False)
    t2 = time.perf_counter()
    inp = np.asarray(coefficients[-1].todense())
    print('level %d: dictionary %s learnt in %.2f s (10 k-means iterations over 10000 windows of %d x %d), encode %.2f s, nnz %d' % (
        level, D.shape, t1 - t0, 2 * w, nfeat, t2 - t1, coefficients[-1].nnz), flush=True)
