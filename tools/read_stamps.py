"""Diagnostic: per-phase cycle shares of the fused greedy loop (libhscmp built with -DHSCMP_DBG_STAMPS)."""
import ctypes, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hsc_amd.synth as synth
from hsc_amd import _native
B = int(os.environ.get('B', '1024'))
_native.LIB_PATH = os.path.abspath(sys.argv[1])
D = synth.make_dictionary(256, 64, seed=2)
x = torch.from_numpy(synth.make_batch(D, 65536, 0, 8, kind='planted', nb_atoms=256, seed=2)).cuda().repeat(B // 8, 1).contiguous()
eng = _native.Engine(0); eng.set_dictionary(D)
params = _native.make_params(nbNonzeroCoefs=256, eps=1.2e-7, maxEvents=576)
lib = _native.load_library()
out = (ctypes.c_ulonglong * 64)()
for i in range(3):
    eng.encode_batch_device(x.data_ptr(), B, 65536, params); eng.synchronize()
    lib.hscmp_debug_stamps(out, 1)
    v = np.array(list(out), dtype=np.float64)
    n = max(v[15], 1)
names = ['dup+update to B1', 'B1', 'energy', 'MFMA tile', 'B4', 'seg+bookkeeping', 'B5', 'deferred stores']
print('B=%d atoms=%d loop %.3f ms' % (B, n, eng.last_kernel_ms()[2]))
for i, nm in enumerate(names):
    print('  %-18s %8.0f cycles/atom' % (nm, v[i] / n))
for i, nm in ((8, 'phase A issue'), (9, 'resolve (Bx..By)'), (10, '[return -> next round]'), (11, '[selection + round checks]'), (12, '[atom body incl. entry]')):
    print('  %-18s %8.0f cycles/atom' % (nm, v[i] / n))
print('  %-18s %8.0f cycles/atom (sum; the select between atoms is not stamped)' % ('total', (v[:8].sum() + v[8] + v[9]) / n))

# per-workgroup residency (diagnostic build)
try:
    nb = min(B, 4096)
    blk = (ctypes.c_ulonglong * (3 * nb))()
    lib.hscmp_debug_blocks(blk, nb)
    a = np.array(list(blk), dtype=np.uint64).reshape(nb, 3)
    st = a[:, 0].astype(np.float64); en = a[:, 1].astype(np.float64)
    t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0      # us (100 MHz)
    dur = en - st
    print('workgroups %d: duration us min %.0f median %.0f max %.0f; last end %.0f us' % (nb, dur.min(), np.median(dur), dur.max(), en.max()))
    for tt in np.linspace(0, en.max(), 12)[:-1]:
        print('  t=%7.0f us resident=%d' % (tt, int(np.sum((st <= tt) & (en > tt)))))
    hw = a[:, 2]
    xcc = (hw >> np.uint64(32)).astype(np.int64); hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf; se = (hwid >> 13) & 0x7; sh_ = (hwid >> 12) & 1
    key = xcc * 1000 + se * 100 + sh_ * 50 + cu
    first = st < 50
    import collections
    cnt = collections.Counter(key[first].tolist())
    print('first-round workgroups: %d on %d distinct CUs; per-CU histogram %s' % (int(first.sum()), len(cnt), dict(collections.Counter(cnt.values()))))
except Exception as ex:
    print('no block info', ex)

# which workgroups share a CU in the first residency round (diagnostic for priority experiments)
try:
    order = np.argsort(st)
    first_idx = np.where(first)[0]
    by_cu = collections.defaultdict(list)
    for i in first_idx:
        by_cu[int(key[i])].append(int(i))
    pairs = [tuple(sorted(v)) for v in by_cu.values() if len(v) == 2]
    diffs = collections.Counter([b - a for a, b in pairs])
    print('pairs sharing a CU: %d; blockIdx difference histogram (top 8): %s' % (len(pairs), diffs.most_common(8)))
    print('parity of (blockIdx >> 8) differs within a pair: %d of %d; parity of blockIdx differs: %d' % (
        sum(((a >> 8) & 1) != ((b >> 8) & 1) for a, b in pairs), len(pairs), sum((a & 1) != (b & 1) for a, b in pairs)))
except Exception as ex:
    print('no pair info', ex)
