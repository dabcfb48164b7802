"""Diagnostic (GPU box host): throughput of bench.py's CPU baseline (NumPy port, one signal per process, BLAS threads = 1)
against the number of worker processes -- the evidence behind the process count the bench uses.  Writes one line per count."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = [int(v) for v in (sys.argv[1:] or ['16', '64', '128', '256'])]
for n in counts:
    payload = dict(B=1024, T=65536, K=256, W=64, L0=256, kind='planted', dtype='f32', nproc=n, per_proc=1)
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-baseline-only', json.dumps(payload)], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    d = json.loads(line[-1]) if line else {'error': out.stderr[-300:]}
    print('%4d processes: %s' % (n, json.dumps({k: d.get(k) for k in ('value', 'per_core', 'cores', 'usable_cores', 'sample', 'error') if k in d})), flush=True)
