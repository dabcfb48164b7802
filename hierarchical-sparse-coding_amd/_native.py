"""ctypes binding of libhscmp.so (C ABI: include/hscmp.h).

There is no CPU implementation behind this module: if the shared library has not been built, or
no MI355X is visible, every entry point raises.  Build with `python __graft_entry__.py build` or
`make -C hierarchical-sparse-coding_amd/csrc`.
"""
import ctypes
import threading
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HSCMP_LIBRARY: another build of the same ABI, for A/B timing of diagnostic builds -- tools/time_lib.py)
LIB_PATH = os.environ.get('HSCMP_LIBRARY') or os.path.join(_HERE, 'csrc', 'libhscmp.so')

F32, F64 = 0, 1
STAT_NNZ, STAT_DUPLICATES, STAT_ROUNDS, STAT_STOP, STAT_ITERATIONS, STAT_EVENTS, STAT_SLOTS, STAT_OFFSET = range(8)
STAT_COUNT = 8
STOP_NAMES = {0: 'running', 1: 'energy_eps', 2: 'nnz', 3: 'snr', 4: 'residual_scale', 5: 'empty',
              6: 'callback', 7: 'capacity', 8: 'stalled', 9: 'group', 100: 'host'}
STOP_RUNNING, STOP_CAPACITY, STOP_STALLED, STOP_GROUP = 0, 7, 8, 9
STOP_HOST = 100      # (host side only: the device loop gave the signal up -- stop reason 'group' -- and the host loop finished it)
METHOD_CMP, METHOD_LOCOMP = 0, 1

# every symbol include/hscmp.h declares (checked by tests/test_abi.py)
EXPORTS = ['hscmp_version', 'hscmp_create', 'hscmp_destroy', 'hscmp_last_error', 'hscmp_set_stream', 'hscmp_set_method',
           'hscmp_synchronize', 'hscmp_set_dictionary', 'hscmp_convolve1d', 'hscmp_select_best_atoms',
           'hscmp_update_inner_products', 'hscmp_table_open', 'hscmp_table_select', 'hscmp_table_update', 'hscmp_table_read', 'hscmp_assign_windows', 'hscmp_host_overlap_add', 'hscmp_host_slots_to_csc', 'hscmp_hierarchy_epilogue', 'hscmp_encode_batch',
           'hscmp_encode_batch_device', 'hscmp_encode_batch_from_level', 'hscmp_continue', 'hscmp_grow_events', 'hscmp_mem_info', 'hscmp_copy_from_device', 'hscmp_stop_signal', 'hscmp_fetch_events',
           'hscmp_fetch_stats', 'hscmp_fetch_residual', 'hscmp_fetch_energies', 'hscmp_fetch_slots',
           'hscmp_get_device_view', 'hscmp_last_kernel_ms', 'hscmp_last_variant']


class HscmpParams(ctypes.Structure):
    _fields_ = [('nb_nonzero_coefs', ctypes.c_int32),
                ('nb_blocks', ctypes.c_int32),
                ('tolerance_snr', ctypes.c_double),
                ('tolerance_residual_scale', ctypes.c_double),
                ('null_coeff_thres', ctypes.c_double),
                ('eps', ctypes.c_double),
                ('max_events', ctypes.c_int32),
                ('max_rounds', ctypes.c_int32)]


class HscmpDeviceView(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int32), ('T', ctypes.c_int32), ('F', ctypes.c_int32), ('K', ctypes.c_int32),
                ('W', ctypes.c_int32), ('max_events', ctypes.c_int32), ('dtype', ctypes.c_int32),
                ('reserved', ctypes.c_int32),
                ('ev_t', ctypes.c_void_p), ('ev_k', ctypes.c_void_p), ('ev_c', ctypes.c_void_p),
                ('stats', ctypes.c_void_p), ('residual', ctypes.c_void_p), ('energies', ctypes.c_void_p),
                ('best_c', ctypes.c_void_p), ('best_k', ctypes.c_void_p)]


class HscmpEpilogueLevel(ctypes.Structure):
    _fields_ = [('col0', ctypes.c_int32), ('col1', ctypes.c_int32), ('scale', ctypes.c_int32), ('rep_is_f32', ctypes.c_int32),
                ('rep', ctypes.c_void_p)]


EVENT_DTYPE = np.dtype('int32,int32,int32,float32')      # hsc/dataset.py:752 (time, level, index, coefficient)


class HscmpError(RuntimeError):
    code = 0            # hscmp_status of the failed call (include/hscmp.h), 0 when raised by the Python layer


ERR_ALLOC = -6
ERR_UNSUPPORTED = -5


_lib = None


def load_library():
    """Load libhscmp.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HscmpError('libhscmp.so is not built (%s). Run `python __graft_entry__.py build` '
                         '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.hscmp_version.restype = ci
    lib.hscmp_create.argtypes = [ctypes.POINTER(vp), ci]
    lib.hscmp_destroy.argtypes = [vp]
    lib.hscmp_destroy.restype = None
    lib.hscmp_last_error.argtypes = [vp]
    lib.hscmp_last_error.restype = ctypes.c_char_p
    lib.hscmp_set_stream.argtypes = [vp, vp]
    lib.hscmp_set_method.argtypes = [vp, ci]
    lib.hscmp_synchronize.argtypes = [vp]
    lib.hscmp_set_dictionary.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    lib.hscmp_convolve1d.argtypes = [vp, vp, ci, ci, vp]
    lib.hscmp_select_best_atoms.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ctypes.c_double, vp, vp, vp, vp, ci,
                                            ctypes.POINTER(ctypes.c_int32)]
    lib.hscmp_update_inner_products.argtypes = [vp, vp, vp, ci, ci]
    lib.hscmp_table_open.argtypes = [vp, vp, ci]
    lib.hscmp_table_select.argtypes = [vp, ci, ci, ctypes.c_double, vp, vp, vp, vp, ci, ctypes.POINTER(ctypes.c_int32)]
    lib.hscmp_table_update.argtypes = [vp, vp, ci, ci, vp, ci]
    lib.hscmp_table_read.argtypes = [vp, vp, vp]
    lib.hscmp_assign_windows.argtypes = [vp, vp, ci, ci, vp, vp, vp]
    lib.hscmp_host_slots_to_csc.argtypes = [vp, vp, vp, ctypes.c_int64, ci, ctypes.c_double, vp, vp, vp]
    lib.hscmp_host_overlap_add.argtypes = [vp, ctypes.c_int64, ci, vp, vp, vp, ctypes.c_int64, vp, ci, ci]
    lib.hscmp_hierarchy_epilogue.argtypes = [vp, vp, ci, ctypes.POINTER(HscmpEpilogueLevel), ci, ctypes.c_double, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.hscmp_encode_batch.argtypes = [vp, vp, ci, ci, ctypes.POINTER(HscmpParams)]
    lib.hscmp_encode_batch_device.argtypes = [vp, vp, ci, ci, ctypes.POINTER(HscmpParams)]
    lib.hscmp_encode_batch_from_level.argtypes = [vp, vp, ci, ci, ctypes.c_double, ctypes.POINTER(HscmpParams)]
    lib.hscmp_continue.argtypes = [vp, ci]
    lib.hscmp_grow_events.argtypes = [vp, ci]
    lib.hscmp_mem_info.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.hscmp_copy_from_device.argtypes = [vp, vp, ctypes.c_uint64, vp]
    lib.hscmp_stop_signal.argtypes = [vp, ci]
    lib.hscmp_fetch_events.argtypes = [vp, vp, vp, vp]
    lib.hscmp_fetch_stats.argtypes = [vp, vp]
    lib.hscmp_fetch_residual.argtypes = [vp, vp]
    lib.hscmp_fetch_energies.argtypes = [vp, vp]
    lib.hscmp_fetch_slots.argtypes = [vp, vp, vp, vp]
    lib.hscmp_get_device_view.argtypes = [vp, ctypes.POINTER(HscmpDeviceView)]
    lib.hscmp_last_kernel_ms.argtypes = [vp, vp]
    lib.hscmp_last_variant.argtypes = [vp]
    lib.hscmp_last_variant.restype = ctypes.c_char_p
    for name in EXPORTS:   # also asserts that every declared symbol is exported
        fn = getattr(lib, name)
        if name not in ('hscmp_destroy', 'hscmp_last_error', 'hscmp_last_variant'):
            fn.restype = ci
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return F32
    if dtype == np.float64:
        return F64
    raise TypeError('the engine computes in float32 or float64, got %s' % dtype)


def nb_blocks_code(nbBlocks):
    if isinstance(nbBlocks, str):
        if nbBlocks != 'auto':
            raise ValueError("nbBlocks must be an integer >= 1 or 'auto'")
        return -1
    nb = int(nbBlocks)
    if nb < 1:
        raise ValueError("nbBlocks must be an integer >= 1 or 'auto'")
    return nb


def max_event_capacity(T):
    """Upper bound of the per-signal event lists: far beyond any converging pursuit (a signal of T samples is
    explained by at most a few atoms per sample), small enough that a non-terminating one fails fast."""
    return max(1 << 18, 16 * int(T))


def make_params(nbNonzeroCoefs=None, toleranceResidualScale=None, toleranceSnr=None, nbBlocks=1,
                minCoefficients=1e-16, eps=None, maxEvents=4096, maxRounds=0):
    nan = float('nan')
    return HscmpParams(
        nb_nonzero_coefs=-1 if nbNonzeroCoefs is None else int(nbNonzeroCoefs),
        nb_blocks=nb_blocks_code(nbBlocks),
        tolerance_snr=nan if toleranceSnr is None else float(toleranceSnr),
        tolerance_residual_scale=nan if toleranceResidualScale is None else float(toleranceResidualScale),
        null_coeff_thres=nan if minCoefficients is None else float(minCoefficients),
        eps=float(eps), max_events=int(maxEvents), max_rounds=int(maxRounds))


class DeviceTable(object):
    """Handle of the inner-product table kept on the device by Engine.table_open (LoCOMP's `innerProducts`).  Quacks
    enough like the reference's ndarray for the coder's own use (shape, dtype); `read()` downloads it."""

    def __init__(self, engine, T):
        self.engine = engine
        self.shape = (T, engine.K)
        self.dtype = engine.dtype
        self._pending = None              # (lo, hi, residual, [centres]): updates not yet sent to the device
        self._generation = engine._table_generation      # the engine holds ONE table: a later table_open retires this handle

    def _check_current(self):
        if self._generation != self.engine._table_generation:
            raise HscmpError('this DeviceTable was replaced by a later table_open (or a new dictionary) on the same engine')

    def defer_update(self, residual, lo, hi, centres):
        """Record an update (residual samples [lo, hi) changed, rows around `centres` to be re-correlated).  Nothing
        reads the table before the next selection, and a row's value only depends on the residual samples of its
        window, all of which are final by then -- so the updates of a selection round go to the device together: the
        changed sample ranges (merged where they touch; NOT their hull -- the atoms of a round are spread over the whole
        signal, and a level >= 1 residual row is F values wide), then all centres in one call."""
        if self._pending is None:
            self._pending = [[(lo, hi)], residual, list(centres)]
        else:
            self._pending[0].append((lo, hi))
            self._pending[1] = residual
            self._pending[2].extend(centres)

    def flush(self):
        self._check_current()
        if self._pending is not None:
            ranges, residual, centres = self._pending
            self._pending = None
            merged = []
            for lo, hi in sorted(ranges):
                if merged and lo <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], hi)
                else:
                    merged.append([lo, hi])
            for lo, hi in merged:
                self.engine.table_update(residual[lo:hi], lo, [])
            self.engine.table_update(residual[0:0], 0, centres)

    def read(self):
        self.flush()
        return self.engine.table_read(table=True)[0]


class Engine(object):
    """One GPU context (hscmp_ctx): resident dictionary + batch workspace."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.hscmp_create(ctypes.byref(h), int(device))
        if rc != 0:
            raise HscmpError('hscmp_create failed (%d): %s' % (rc, self._lib.hscmp_last_error(None).decode()))
        self._h = h
        self.device = int(device)
        self.dtype = None
        self.K = self.W = self.F = None
        self._batch = None

    def close(self):
        if getattr(self, '_h', None):
            self._lib.hscmp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            ex = HscmpError('%s failed (%d): %s' % (what, rc, self._lib.hscmp_last_error(self._h).decode()))
            ex.code = int(rc)
            raise ex

    def set_stream(self, stream_ptr):
        self._check(self._lib.hscmp_set_stream(self._h, ctypes.c_void_p(stream_ptr or 0)), 'hscmp_set_stream')

    def set_method(self, method):
        """METHOD_CMP (default) or METHOD_LOCOMP for the batch encodes that follow (sticky: reset it when done -- engines are shared)."""
        self._check(self._lib.hscmp_set_method(self._h, int(method)), 'hscmp_set_method')

    def synchronize(self):
        self._check(self._lib.hscmp_synchronize(self._h), 'hscmp_synchronize')

    def set_dictionary(self, D, weights=None, dtype=None):
        """D [K,W] or [K,W,F] float32/float64 (C order); weights [K] or None.  dtype: convert D (and the weights) to this
        type for the upload -- the identity check runs on the caller's own array first, so an unchanged dictionary costs
        one checksum of the source and no conversion (a level dictionary of BASELINE config 5 is 70 MB as float64)."""
        assert D.ndim in (2, 3)
        if dtype is not None and D.dtype != np.dtype(dtype):
            src = np.ascontiguousarray(D)
            skey = (_dictionary_key(src.reshape((src.shape[0], src.shape[1], -1)), None if weights is None else np.ascontiguousarray(weights)), np.dtype(dtype).str)
            if skey == getattr(self, '_src_key', None) and getattr(self, '_dict_key', None) is not None:
                return
            self._src_key = None
            self.set_dictionary(np.ascontiguousarray(D, dtype=dtype), None if weights is None else np.asarray(weights, dtype=dtype))
            self._src_key = skey
            return
        self._src_key = None
        D3 = np.ascontiguousarray(D.reshape((D.shape[0], D.shape[1], -1)))
        code = dtype_code(D3.dtype)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=D3.dtype)
        if w is not None and w.shape != (D3.shape[0],):
            raise ValueError('weights must have one entry per atom')
        K, W, F = D3.shape
        # the same dictionary again (the row-level hooks of LoCOMP-style callers pass it with every call): nothing to upload
        key = _dictionary_key(D3, w)
        if key == getattr(self, '_dict_key', None):
            return
        self._dict_key = None
        self._check(self._lib.hscmp_set_dictionary(self._h, _ptr(D3), K, W, F, code, _ptr(w)), 'hscmp_set_dictionary')
        self._dict_key = key
        self.dtype, self.K, self.W, self.F = D3.dtype, K, W, F
        self._batch = None
        # (a table opened under the previous dictionary is retired: engine_for recycles its least recently used engine, and a
        #  handle that outlived that must fail loudly instead of reading a table of another dictionary)
        self._table_generation = getattr(self, '_table_generation', 0) + 1

    def convolve1d(self, x, same):
        x2 = np.ascontiguousarray(x.reshape((x.shape[0], -1)), dtype=self.dtype)
        assert x2.shape[1] == self.F
        T = x2.shape[0]
        Tout = T if same else T - self.W + 1
        if Tout <= 0:
            raise HscmpError('sequence shorter than the filters')
        out = np.empty((Tout, self.K), dtype=self.dtype)
        self._check(self._lib.hscmp_convolve1d(self._h, _ptr(x2), T, 1 if same else 0, _ptr(out)), 'hscmp_convolve1d')
        return out

    def assign_windows(self, windows):
        """modeling.py:454-460: per window [L,F] of `windows` [N,L(,F)] the (position, atom, coefficient) of the
        largest |valid correlation| with the dictionary, ties in C order."""
        w3 = np.ascontiguousarray(windows.reshape((windows.shape[0], windows.shape[1], -1)), dtype=self.dtype)
        assert w3.shape[2] == self.F
        N, L = w3.shape[0], w3.shape[1]
        t = np.empty((N,), dtype=np.int32); k = np.empty((N,), dtype=np.int32); c = np.empty((N,), dtype=self.dtype)
        self._check(self._lib.hscmp_assign_windows(self._h, _ptr(w3), N, L, _ptr(t), _ptr(k), _ptr(c)), 'hscmp_assign_windows')
        return t, k, c

    def select_best_atoms(self, innerProducts, filterWidth, nbBlocks=1, offset=False, nullCoeffThres=0.0, weights=None):
        """modeling.py:899-982 on a materialised table [T,K]; returns (t, k, c) in the reference's order."""
        ip = np.ascontiguousarray(innerProducts)
        code = dtype_code(ip.dtype)
        T, K = ip.shape
        w = None if weights is None else np.ascontiguousarray(weights, dtype=ip.dtype)
        cap = T + 2
        t = np.empty(cap, dtype=np.int32)
        k = np.empty(cap, dtype=np.int32)
        c = np.empty(cap, dtype=ip.dtype)
        n = ctypes.c_int32(0)
        thres = float('nan') if nullCoeffThres is None else float(nullCoeffThres)
        self._check(self._lib.hscmp_select_best_atoms(self._h, _ptr(ip), T, K, int(filterWidth), code, nb_blocks_code(nbBlocks),
                                                      int(bool(offset)), ctypes.c_double(thres), _ptr(w), _ptr(t), _ptr(k), _ptr(c),
                                                      cap, ctypes.byref(n)), 'hscmp_select_best_atoms')
        self._batch = None
        return t[:n.value].copy(), k[:n.value].copy(), c[:n.value].copy()

    def update_inner_products(self, innerProducts, residual, position):
        """modeling.py:1018-1051 for one atom centre, in place on innerProducts [T,K] (dictionary of the context)."""
        assert innerProducts.flags.c_contiguous and innerProducts.dtype == self.dtype
        r = np.ascontiguousarray(residual.reshape((residual.shape[0], -1)), dtype=self.dtype)
        assert r.shape[1] == self.F and innerProducts.shape == (r.shape[0], self.K)
        self._check(self._lib.hscmp_update_inner_products(self._h, _ptr(innerProducts), _ptr(r), r.shape[0], int(position)),
                    'hscmp_update_inner_products')
        return innerProducts

    # ---- LoCOMP's table, resident on the device (include/hscmp.h: hscmp_table_*) ----
    def table_open(self, x):
        """innerProducts = convolve1d(x, D, 'same') and residual := x, both kept on the device (modeling.py:1293)."""
        r = np.ascontiguousarray(np.asarray(x).reshape((np.asarray(x).shape[0], -1)), dtype=self.dtype)
        assert r.shape[1] == self.F
        self._check(self._lib.hscmp_table_open(self._h, _ptr(r), r.shape[0]), 'hscmp_table_open')
        self._table_T = r.shape[0]
        self._table_generation = getattr(self, '_table_generation', 0) + 1
        return DeviceTable(self, r.shape[0])

    def table_select(self, nbBlocks=1, offset=False, nullCoeffThres=0.0, weights=None):
        T = self._table_T
        w = None if weights is None else np.ascontiguousarray(weights, dtype=self.dtype)
        cap = T + 2
        t = np.empty(cap, dtype=np.int32)
        k = np.empty(cap, dtype=np.int32)
        c = np.empty(cap, dtype=self.dtype)
        n = ctypes.c_int32(0)
        thres = float('nan') if nullCoeffThres is None else float(nullCoeffThres)
        self._check(self._lib.hscmp_table_select(self._h, nb_blocks_code(nbBlocks), int(bool(offset)), ctypes.c_double(thres), _ptr(w),
                                                 _ptr(t), _ptr(k), _ptr(c), cap, ctypes.byref(n)), 'hscmp_table_select')
        self._batch = None
        return t[:n.value].copy(), k[:n.value].copy(), c[:n.value].copy()

    def table_update(self, residual_rows, start, centres):
        """residual[start:start+len(rows)] := rows on the device, then rows p-(W-1)..p+(W-1) of the table re-correlated
        in place for every centre p (modeling.py:1018-1051)."""
        if len(residual_rows) == 0:
            rows = np.zeros((0, self.F), dtype=self.dtype)
        else:
            rows = np.ascontiguousarray(np.asarray(residual_rows).reshape((len(residual_rows), -1)), dtype=self.dtype)
        cs = np.ascontiguousarray(centres, dtype=np.int32)
        self._check(self._lib.hscmp_table_update(self._h, _ptr(rows), int(start), rows.shape[0], _ptr(cs), cs.shape[0]), 'hscmp_table_update')

    def table_read(self, table=True, residual=False):
        T = self._table_T
        tab = np.empty((T, self.K), dtype=self.dtype) if table else None
        res = np.empty((T, self.F), dtype=self.dtype) if residual else None
        self._check(self._lib.hscmp_table_read(self._h, _ptr(tab), _ptr(res)), 'hscmp_table_read')
        return tab, res

    def encode_batch(self, x, params):
        """x [B,T,F] host array of the dictionary dtype."""
        x3 = np.ascontiguousarray(x, dtype=self.dtype)
        assert x3.ndim == 3 and x3.shape[2] == self.F
        B, T = x3.shape[0], x3.shape[1]
        self._check(self._lib.hscmp_encode_batch(self._h, _ptr(x3), B, T, ctypes.byref(params)), 'hscmp_encode_batch')
        self._batch = (B, T, int(params.max_events))

    def encode_batch_device(self, x_dev_ptr, B, T, params):
        """x_dev_ptr: device address of [B,T,F] in the dictionary dtype; asynchronous."""
        self._check(self._lib.hscmp_encode_batch_device(self._h, ctypes.c_void_p(x_dev_ptr), int(B), int(T),
                                                        ctypes.byref(params)), 'hscmp_encode_batch_device')
        self._batch = (int(B), int(T), int(params.max_events))

    def encode_batch_from_level(self, prev, first, count, minCoefficients, params):
        """Hierarchical level chaining on the device (modeling.py:1489): the coefficient slots of signals
        [first, first+count) of engine `prev` become this engine's dense float64 input."""
        minc = float('nan') if minCoefficients is None else float(minCoefficients)
        self._check(self._lib.hscmp_encode_batch_from_level(self._h, prev._h, int(first), int(count), ctypes.c_double(minc),
                                                            ctypes.byref(params)), 'hscmp_encode_batch_from_level')
        self._batch = (int(count), prev._batch[1], int(params.max_events))

    def hierarchy_epilogue(self, level0, first, levels, minCoefficients, slot_counts, want_events=True, want_residual=True,
                           residual_out=None, energy_out=None):
        """hscmp_hierarchy_epilogue on this (last-level) engine.  levels: list of (col0, col1, representations [K,scale(,Fd)]).
        slot_counts: int array [count], the slot count of every signal (stats[:, STAT_SLOTS]).
        energy_out: float64 [count] array that receives the residual energies (sum of squares, summed on the device).
        Returns (n [count], colptr [count, K+1], offsets [count+1], indices, data, events or None, residual or None)."""
        count, T, _ = self._batch
        Fd = level0.F
        arr = (HscmpEpilogueLevel * len(levels))()
        keep = []
        for i, (c0, c1, rep) in enumerate(levels):
            arr[i].col0, arr[i].col1 = int(c0), int(c1)
            if rep is None or c1 <= c0:
                arr[i].scale, arr[i].rep_is_f32, arr[i].rep = 0, 1, None
                continue
            r = np.ascontiguousarray(rep.reshape((rep.shape[0], rep.shape[1], -1)))
            if r.dtype not in (np.float32, np.float64):
                r = r.astype(np.float64)
            assert r.shape[0] >= c1 and r.shape[2] == Fd
            keep.append(r)
            arr[i].scale, arr[i].rep_is_f32, arr[i].rep = int(r.shape[1]), 1 if r.dtype == np.float32 else 0, r.ctypes.data
        offsets = np.zeros(count + 1, dtype=np.int64)
        np.cumsum(np.asarray(slot_counts, dtype=np.int64), out=offsets[1:])
        total = int(offsets[-1])
        n = np.empty(count, dtype=np.int32)
        colptr = np.empty((count, self.K + 1), dtype=np.int32)
        indices = np.empty(max(total, 1), dtype=np.int32)
        data = np.empty(max(total, 1), dtype=np.float64)
        events = np.empty(max(total, 1), dtype=EVENT_DTYPE) if want_events else None
        if residual_out is not None:                 # the caller's [count, T, Fd] float64 slice (C-contiguous) is filled in place
            assert residual_out.shape == (count, T, Fd) and residual_out.dtype == np.float64 and residual_out.flags.c_contiguous
            residual = residual_out
        else:
            residual = np.empty((count, T, Fd), dtype=np.float64) if want_residual else None
        if energy_out is not None:
            assert energy_out.shape == (count,) and energy_out.dtype == np.float64 and energy_out.flags.c_contiguous
        minc = float('nan') if minCoefficients is None else float(minCoefficients)
        self._check(self._lib.hscmp_hierarchy_epilogue(self._h, level0._h, int(first), arr, len(levels), ctypes.c_double(minc), _ptr(offsets),
                                                       _ptr(n), _ptr(colptr), _ptr(indices), _ptr(data), _ptr(events), _ptr(residual),
                                                       _ptr(energy_out)),
                    'hscmp_hierarchy_epilogue')
        return n, colptr, offsets, indices, data, events, residual

    def continue_rounds(self, max_rounds):
        self._check(self._lib.hscmp_continue(self._h, int(max_rounds)), 'hscmp_continue')

    def grow_events(self, max_events):
        """Larger event / slot lists, contents kept; signals stopped on capacity run again on continue_rounds()."""
        self._check(self._lib.hscmp_grow_events(self._h, int(max_events)), 'hscmp_grow_events')
        self._batch = (self._batch[0], self._batch[1], int(max_events))

    def mem_info(self):
        """(free, total) bytes of this engine's GPU."""
        f, t = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.hscmp_mem_info(self._h, ctypes.byref(f), ctypes.byref(t)), 'hscmp_mem_info')
        return int(f.value), int(t.value)

    def copy_from_device(self, dev_ptr, shape, dtype):
        """Device memory at `dev_ptr` (this engine's GPU) as a new host array of the given shape / dtype."""
        out = np.empty(shape, dtype=dtype)
        self._check(self._lib.hscmp_copy_from_device(self._h, ctypes.c_void_p(int(dev_ptr)), ctypes.c_uint64(out.nbytes), _ptr(out)), 'hscmp_copy_from_device')
        return out

    def stop_signal(self, b):
        self._check(self._lib.hscmp_stop_signal(self._h, int(b)), 'hscmp_stop_signal')

    def fetch_stats(self):
        B = self._batch[0]
        st = np.empty((B, STAT_COUNT), dtype=np.int32)
        self._check(self._lib.hscmp_fetch_stats(self._h, _ptr(st)), 'hscmp_fetch_stats')
        return st

    def fetch_events(self):
        B, _, cap = self._batch
        t = np.empty((B, cap), dtype=np.int32)
        k = np.empty((B, cap), dtype=np.int32)
        c = np.empty((B, cap), dtype=self.dtype)
        self._check(self._lib.hscmp_fetch_events(self._h, _ptr(t), _ptr(k), _ptr(c)), 'hscmp_fetch_events')
        return t, k, c

    # raw-pointer forms (pinned host buffers owned by the caller, e.g. bench.py's PCIe-inclusive leg)
    def fetch_events_into(self, t_ptr, k_ptr, c_ptr):
        self._check(self._lib.hscmp_fetch_events(self._h, ctypes.c_void_p(t_ptr), ctypes.c_void_p(k_ptr), ctypes.c_void_p(c_ptr)), 'hscmp_fetch_events')

    def fetch_stats_into(self, ptr):
        self._check(self._lib.hscmp_fetch_stats(self._h, ctypes.c_void_p(ptr)), 'hscmp_fetch_stats')

    def fetch_energies_into(self, ptr):
        self._check(self._lib.hscmp_fetch_energies(self._h, ctypes.c_void_p(ptr)), 'hscmp_fetch_energies')

    def fetch_slots(self):
        B, _, cap = self._batch
        t = np.empty((B, cap), dtype=np.int32)
        k = np.empty((B, cap), dtype=np.int32)
        a = np.empty((B, cap), dtype=np.float64)
        self._check(self._lib.hscmp_fetch_slots(self._h, _ptr(t), _ptr(k), _ptr(a)), 'hscmp_fetch_slots')
        return t, k, a

    def fetch_residual(self):
        B, T, _ = self._batch
        r = np.empty((B, T, self.F), dtype=self.dtype)
        self._check(self._lib.hscmp_fetch_residual(self._h, _ptr(r)), 'hscmp_fetch_residual')
        return r

    def fetch_energies(self):
        B = self._batch[0]
        e = np.empty((B, 2), dtype=np.float64)
        self._check(self._lib.hscmp_fetch_energies(self._h, _ptr(e)), 'hscmp_fetch_energies')
        return e

    def device_view(self):
        v = HscmpDeviceView()
        self._check(self._lib.hscmp_get_device_view(self._h, ctypes.byref(v)), 'hscmp_get_device_view')
        return v

    def last_kernel_ms(self):
        out = np.zeros(4, dtype=np.float32)
        self._check(self._lib.hscmp_last_kernel_ms(self._h, _ptr(out)), 'hscmp_last_kernel_ms')
        return out

    def last_variant(self):
        return self._lib.hscmp_last_variant(self._h).decode()


def _dictionary_key(D3, w):
    """Identity of an uploaded dictionary: shape, dtype, checksums of the CONTENTS (a learner updates its dictionary in
    place), and the diagnostic switches read at upload."""
    return (D3.shape, D3.dtype.str, _checksum(D3), None if w is None else _checksum(w),
            tuple(sorted(kv for kv in os.environ.items() if kv[0].startswith('HSCMP_'))))


def _checksum(a):
    """Checksum of an array's bytes.  A level dictionary of BASELINE config 5 is 70 MB and is looked up once per batch and
    level: xxh3 (when the module is there) reads it at memory speed, crc32 takes 40 ms."""
    buf = a.view(np.uint8).reshape(-1)
    try:
        import xxhash
        one = xxhash.xxh3_64_intdigest
    except ImportError:
        import zlib
        one = zlib.crc32
    if buf.size < (8 << 20):
        return one(buf)
    # large arrays: 4 MB pieces on a few threads (both hashes release the interpreter lock on large buffers)
    global _hash_pool
    if _hash_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _hash_pool = ThreadPoolExecutor(max_workers=8)
    step = 4 << 20
    return tuple(_hash_pool.map(one, [buf[i:i + step] for i in range(0, buf.size, step)]))


_hash_pool = None


_engines = threading.local()
kEnginesPerThread = 4


def default_engine(device=0):
    """One engine per (thread, device): a context is single-threaded (include/hscmp.h), so every thread of the host
    program gets its own -- the coders of hsc_amd.modeling can then run side by side in a thread pool (the ctypes calls
    release the interpreter lock; the contexts use separate HIP streams).  The dictionary is re-uploaded when it changes."""
    table = _engines.__dict__.setdefault('by_device', {})
    if device not in table:
        table[device] = Engine(device)
    return table[device]


def engine_for(device, D, weights=None):
    """An engine of this thread that already HOLDS the dictionary (uploaded, lists built), else the least recently used of
    up to kEnginesPerThread engines per device, with the dictionary set.  A hierarchical encode alternates between its
    level dictionaries signal after signal: with one engine every level would upload its dictionary again each time
    (13 MB and ~0.1 s of list building for config-4 level 1)."""
    D3 = np.ascontiguousarray(D.reshape((D.shape[0], D.shape[1], -1)))
    w = None if weights is None else np.ascontiguousarray(weights, dtype=D3.dtype)
    key = _dictionary_key(D3, w)
    pool = _engines.__dict__.setdefault('pool', {}).setdefault(device, [])
    for i, eng in enumerate(pool):
        if getattr(eng, '_dict_key', None) == key:
            pool.append(pool.pop(i))                      # most recently used last
            return eng
    if len(pool) < kEnginesPerThread:
        eng = Engine(device)
    else:
        eng = pool.pop(0)
    pool.append(eng)
    eng.set_dictionary(D3, w)
    return eng
