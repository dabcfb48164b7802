"""The host side of the C-ABI shim under AddressSanitizer + UBSan (CPU only; `make asan` in csrc/): a child interpreter
with the sanitizer runtime preloaded loads libhscmp_asan.so and runs tests/asan_child.py -- NULL / bad arguments to
every entry point, the no-device path of hscmp_create, and the host-side helpers on randomized inputs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'hierarchical-sparse-coding_amd', 'csrc')


def test_host_shim_under_asan_ubsan():
    lib = os.path.join(CSRC, 'libhscmp_asan.so')
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))] + [os.path.join(ROOT, 'include', 'hscmp.h')]
    if not os.path.isfile(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(['make', '-C', CSRC, 'asan'], check=True, capture_output=True, timeout=900)
    rt = subprocess.run(['make', '-s', '-C', CSRC, 'asan-runtime'], check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    if not os.path.isfile(rt):
        pytest.skip('no AddressSanitizer runtime next to hipcc (%s)' % rt)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=23', UBSAN_OPTIONS='halt_on_error=1:exitcode=24',
               HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'asan_child.py'), lib], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'ASAN-CHILD-OK' in res.stdout, (res.returncode, res.stdout[-2000:], res.stderr[-4000:])
