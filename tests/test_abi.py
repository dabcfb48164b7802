"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/hscmp.h declares (no compute calls here -- those are the -m gpu tests)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'hscmp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(hscmp_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from hsc_amd import _native
    if not os.path.isfile(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), 'libhscmp.so does not export %s' % name
    assert sorted(_native.EXPORTS) == declared
    lib.hscmp_version.restype = ctypes.c_int
    assert lib.hscmp_version() == 100


def test_no_gpu_fails_loudly():
    """Without a GPU the engine must raise, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    import numpy as np
    from hsc_amd import _native
    from hsc_amd.modeling import ConvolutionalMatchingPursuit
    with pytest.raises(_native.HscmpError):
        _native.Engine(0)
    with pytest.raises(_native.HscmpError):
        ConvolutionalMatchingPursuit().computeCoefficients(np.zeros(64, dtype=np.float32),
                                                           np.ones((2, 4), dtype=np.float32), nbNonzeroCoefs=1)


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under the package may import or link it."""
    pkg = os.path.join(ROOT, 'hierarchical-sparse-coding_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')) or f == 'Makefile':
                text = open(os.path.join(dirpath, f)).read()
                for needle in ('hsc_oracle', 'from oracle', 'import oracle', 'libhsc_oracle'):
                    assert needle not in text, '%s references the oracle (%s)' % (f, needle)
