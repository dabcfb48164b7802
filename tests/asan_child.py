"""Child process of tests/test_abi_asan.py: runs with the AddressSanitizer runtime preloaded, loads the sanitized build of
the shim (csrc/libhscmp_asan.so) and drives every entry point that works without a GPU: argument checks, error
strings, and the two host-side helpers on randomized inputs (checked against plain numpy).  Any sanitizer report ends
the process with a non-zero status."""
import ctypes
import sys

import numpy as np

lib = ctypes.CDLL(sys.argv[1])
vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.hscmp_last_error.restype = ctypes.c_char_p
lib.hscmp_last_error.argtypes = [vp]
assert lib.hscmp_version() == 100

# no device: creation fails with a message, never a context
h = vp()
rc = lib.hscmp_create(ctypes.byref(h), 0)
assert rc != 0 and not h.value, rc
assert len(lib.hscmp_last_error(None)) > 0
assert lib.hscmp_create(None, 0) != 0

# every context entry point rejects a NULL context without touching its other arguments
null = vp(None)
buf = (ctypes.c_char * 64)()
lib.hscmp_set_dictionary.argtypes = [vp, vp, ci, ci, ci, ci, vp]
assert lib.hscmp_set_dictionary(null, buf, 1, 1, 1, 0, None) != 0
lib.hscmp_convolve1d.argtypes = [vp, vp, ci, ci, vp]
assert lib.hscmp_convolve1d(null, buf, 4, 1, buf) != 0
lib.hscmp_select_best_atoms.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ctypes.c_double, vp, vp, vp, vp, ci, vp]
assert lib.hscmp_select_best_atoms(null, buf, 4, 1, 1, 0, 1, 0, 0.0, None, buf, buf, buf, 1, buf) != 0
lib.hscmp_update_inner_products.argtypes = [vp, vp, vp, ci, ci]
assert lib.hscmp_update_inner_products(null, buf, buf, 4, 0) != 0
lib.hscmp_table_open.argtypes = [vp, vp, ci]
assert lib.hscmp_table_open(null, buf, 4) != 0
lib.hscmp_table_select.argtypes = [vp, ci, ci, ctypes.c_double, vp, vp, vp, vp, ci, vp]
assert lib.hscmp_table_select(null, 1, 0, 0.0, None, buf, buf, buf, 1, buf) != 0
lib.hscmp_table_update.argtypes = [vp, vp, ci, ci, vp, ci]
assert lib.hscmp_table_update(null, buf, 0, 1, buf, 1) != 0
lib.hscmp_table_read.argtypes = [vp, vp, vp]
assert lib.hscmp_table_read(null, buf, None) != 0
lib.hscmp_assign_windows.argtypes = [vp, vp, ci, ci, vp, vp, vp]
assert lib.hscmp_assign_windows(null, buf, 1, 4, buf, buf, buf) != 0
lib.hscmp_encode_batch.argtypes = [vp, vp, ci, ci, vp]
assert lib.hscmp_encode_batch(null, buf, 1, 4, buf) != 0
lib.hscmp_encode_batch_device.argtypes = [vp, vp, ci, ci, vp]
assert lib.hscmp_encode_batch_device(null, buf, 1, 4, buf) != 0
lib.hscmp_encode_batch_from_level.argtypes = [vp, vp, ci, ci, ctypes.c_double, vp]
assert lib.hscmp_encode_batch_from_level(null, null, 0, 1, 0.0, buf) != 0
lib.hscmp_continue.argtypes = [vp, ci]
assert lib.hscmp_continue(null, 1) != 0
lib.hscmp_grow_events.argtypes = [vp, ci]
assert lib.hscmp_grow_events(null, 8) != 0
lib.hscmp_stop_signal.argtypes = [vp, ci]
assert lib.hscmp_stop_signal(null, 0) != 0
lib.hscmp_mem_info.argtypes = [vp, vp, vp]
assert lib.hscmp_mem_info(null, buf, buf) != 0
for name in ('hscmp_fetch_stats', 'hscmp_fetch_residual', 'hscmp_fetch_energies', 'hscmp_last_kernel_ms', 'hscmp_get_device_view'):
    fn = getattr(lib, name)
    fn.argtypes = [vp, vp]
    assert fn(null, buf) != 0, name
lib.hscmp_fetch_events.argtypes = [vp, vp, vp, vp]
assert lib.hscmp_fetch_events(null, buf, buf, buf) != 0
lib.hscmp_fetch_slots.argtypes = [vp, vp, vp, vp]
assert lib.hscmp_fetch_slots(null, buf, buf, buf) != 0
lib.hscmp_hierarchy_epilogue.argtypes = [vp, vp, ci, vp, ci, ctypes.c_double, vp, vp, vp, vp, vp, vp, vp, vp]
assert lib.hscmp_hierarchy_epilogue(null, null, 0, buf, 1, 0.0, buf, buf, buf, buf, buf, None, None, None) != 0
lib.hscmp_synchronize.argtypes = [vp]
assert lib.hscmp_synchronize(null) != 0
lib.hscmp_set_stream.argtypes = [vp, vp]
assert lib.hscmp_set_stream(null, None) != 0
lib.hscmp_last_variant.argtypes = [vp]
lib.hscmp_last_variant.restype = ctypes.c_char_p
assert lib.hscmp_last_variant(null) == b''
lib.hscmp_destroy.argtypes = [vp]
lib.hscmp_destroy(null)


def ptr(a):
    return a.ctypes.data_as(vp)


# host helpers on random inputs (no GPU, no context): results against numpy
lib.hscmp_host_slots_to_csc.argtypes = [vp, vp, vp, i64, ci, ctypes.c_double, vp, vp, vp]
lib.hscmp_host_overlap_add.argtypes = [vp, i64, ci, vp, vp, vp, i64, vp, ci, ci]
rs = np.random.RandomState(5)
for case in range(40):
    K, T = int(rs.randint(1, 9)), int(rs.randint(4, 60))
    n = int(rs.randint(0, 50))
    pairs = rs.permutation(K * T)[:min(n, K * T)]
    n = len(pairs)
    st, sk = (pairs // K).astype(np.int32), (pairs % K).astype(np.int32)
    sa = rs.standard_normal(n)
    sa[rs.rand(n) < 0.2] = 0.0
    minc = [float('nan'), 0.3][case % 2]
    indptr = np.zeros(K + 1, dtype=np.int32)
    indices = np.zeros(max(n, 1), dtype=np.int32)
    data = np.zeros(max(n, 1), dtype=np.float64)
    assert lib.hscmp_host_slots_to_csc(ptr(st), ptr(sk), ptr(sa), n, K, minc, ptr(indptr), ptr(indices), ptr(data)) == 0
    keep = (sa != 0.0) & ((np.abs(sa) >= minc) if minc == minc else True)
    order = np.lexsort((st[keep], sk[keep]))
    assert indptr[K] == keep.sum()
    assert np.array_equal(indices[:indptr[K]], st[keep][order]) and np.array_equal(data[:indptr[K]], sa[keep][order])
    assert np.array_equal(indptr, np.concatenate([[0], np.cumsum(np.bincount(sk[keep], minlength=K))]))
    # overlap-add of those entries (centred placement, clipping at both ends: utils.py:84-131)
    W, Fd = int(rs.randint(1, 12)), int(rs.randint(1, 4))
    f32 = bool(case & 2)
    Dd = rs.standard_normal((K, W, Fd)).astype(np.float32 if f32 else np.float64)
    rows, cols = st[keep][order].astype(np.int64), sk[keep][order].astype(np.int64)
    vals = sa[keep][order].copy()
    sig = np.zeros((T, Fd))
    assert lib.hscmp_host_overlap_add(ptr(sig), T, Fd, ptr(rows), ptr(cols), ptr(vals), len(vals), ptr(Dd), W, int(f32)) == 0
    exp = np.zeros((T, Fd))
    for t, k, c in zip(rows, cols, vals):
        lo = int(t) - (W - 1) // 2
        s, e = max(lo, 0), min(lo + W, T)
        exp[s:e] += c * Dd[k][s - lo:e - lo].astype(np.float64)
    assert np.array_equal(sig, exp), case
# and their argument checks
assert lib.hscmp_host_slots_to_csc(None, None, None, 3, 2, 0.0, None, None, None) != 0
assert lib.hscmp_host_overlap_add(None, 4, 1, None, None, None, 2, None, 3, 0) != 0
print('ASAN-CHILD-OK')
