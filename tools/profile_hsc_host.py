"""Diagnostic (GPU box): host-side profile of one hierarchical batch encode at the config-4 shape (what is NOT kernel time)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_hsc
from hsc_amd.hierarchical import HierarchicalConvolutionalMatchingPursuit

B = int(os.environ.get('B', '1024'))
mlds, xs, kw, desc = bench_hsc.build_workload(int(os.environ.get('CONFIG', '4')), B, 65536, 0, 17)
x_dev = torch.from_numpy(xs).cuda()
h = HierarchicalConvolutionalMatchingPursuit(method='cmp')
for _ in range(2):
    t0 = time.perf_counter()
    coefs, res, tim = h.computeCoefficientsBatch(xs, mlds, deviceInput=x_dev.data_ptr(), residuals='energy', **kw)
    print('wall %.1f ms, kernels %.1f ms' % (1e3 * (time.perf_counter() - t0), sum(sum(t['kernel_ms'][:3]) for t in tim)), flush=True)
cProfile.run("h.computeCoefficientsBatch(xs, mlds, deviceInput=x_dev.data_ptr(), residuals='energy', **kw)", '/tmp/hsc.prof')
pstats.Stats('/tmp/hsc.prof').sort_stats('cumulative').print_stats(22)
