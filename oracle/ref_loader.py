"""TEST INFRASTRUCTURE ONLY -- loader for the Python-2 reference in THIS container.

Loads `/root/reference/hsc/{utils,analysis,dataset,modeling}.py` (Python 2.7 sources,
SURVEY.md section 8c) into synthetic in-memory modules so that the *real* reference can be
run to (i) validate the restatements under `oracle/` and (ii) emit the golden vectors under
`tests/golden/` (script: `tools/make_golden.py`).

Nothing is written to `/root/reference`, nothing of the reference is copied into this repo:
the sources are read as text, a handful of regex patches (py2 -> py3 spelling only) are applied
in memory and the result is `exec`-ed.  When `/root/reference` is absent (the GPU box) the
loader returns None -- no test that runs there may depend on it.

Patches (all semantic no-ops for the Python-2 meaning of the code):
  * `cPickle` -> alias of `pickle`, `StringIO` module stub, `itertools.izip` -> `zip`,
    `collections.Iterable` -> `collections.abc.Iterable`, `np.int`/`np.float` -> `int`/`float`
  * int `/2` -> `//2` (Python-2 floor division on ints), except the float `bar_width/2`
    sites in analysis.py
  * `np.unravel_index(dims=` -> `shape=`, `np.Inf` -> `np.inf`,
    `np.issubdtype(x, np.float|np.int)` -> `np.floating|np.integer`
"""
import os
import re
import sys
import types

REFERENCE_ROOT = '/root/reference'

_cache = {}


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'hsc', 'modeling.py'))


def _patch_source(name, text):
    if name == 'analysis':
        text = text.replace('bar_width/2', 'bar_width/2.0')
    text = re.sub(r'/2(?![.\d])', '//2', text)
    text = text.replace('dims=(', 'shape=(')
    text = text.replace('np.Inf', 'np.inf')
    text = re.sub(r'np\.issubdtype\(([^,]+),\s*np\.float\)', r'np.issubdtype(\1, np.floating)', text)
    text = re.sub(r'np\.issubdtype\(([^,]+),\s*np\.int\)', r'np.issubdtype(\1, np.integer)', text)
    return text


def load_reference():
    """Returns a namespace object with attributes utils, analysis, dataset, modeling, or None."""
    if 'ns' in _cache:
        return _cache['ns']
    if not reference_available():
        _cache['ns'] = None
        return None

    import collections
    import collections.abc
    import io
    import itertools
    import pickle

    import numpy as np

    # py2 module / attribute spellings
    sys.modules.setdefault('cPickle', pickle)
    if 'StringIO' not in sys.modules:
        sio = types.ModuleType('StringIO')
        sio.StringIO = io.StringIO
        sys.modules['StringIO'] = sio
    if not hasattr(itertools, 'izip'):
        itertools.izip = zip
    if not hasattr(collections, 'Iterable'):
        collections.Iterable = collections.abc.Iterable
    if not hasattr(np, 'int'):
        np.int = int
    if not hasattr(np, 'float'):
        np.float = float

    import matplotlib
    matplotlib.use('Agg')

    pkg = types.ModuleType('hsc')
    pkg.__path__ = []
    saved = {k: sys.modules.get(k) for k in ('hsc', 'hsc.utils', 'hsc.analysis', 'hsc.dataset', 'hsc.modeling')}
    sys.modules['hsc'] = pkg
    mods = {}
    try:
        for name in ('utils', 'analysis', 'dataset', 'modeling'):
            path = os.path.join(REFERENCE_ROOT, 'hsc', name + '.py')
            with open(path, 'r') as f:
                text = _patch_source(name, f.read())
            mod = types.ModuleType('hsc.' + name)
            mod.__file__ = path
            sys.modules['hsc.' + name] = mod
            setattr(pkg, name, mod)
            exec(compile(text, path, 'exec'), mod.__dict__)
            mods[name] = mod
    finally:
        # do not leave a fake top-level `hsc` package behind for unrelated importers
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    ns = types.SimpleNamespace(**mods)
    _cache['ns'] = ns
    return ns
