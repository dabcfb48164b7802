"""Build-container helper (tools/ only): run the REAL reference (via oracle/ref_loader.py) and
capture the ordered (t, k, c) selection trace of ConvolutionalMatchingPursuit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle.ref_loader import load_reference  # noqa: E402


def traced_cmp(ref):
    """Subclass of the reference CMP that logs every atom it applies (hook: _updateCoefficients,
    modeling.py:984, called once per applied atom at :1114)."""
    base = ref.modeling.ConvolutionalMatchingPursuit

    class TracedCMP(base):
        def __init__(self):
            base.__init__(self, verbose=False)
            self.trace = []

        def _updateCoefficients(self, coefficients, atoms, replace=True):
            for a in atoms:
                self.trace.append((int(a.position), int(a.index), a.coefficient))
            return base._updateCoefficients(self, coefficients, atoms, replace)

    return TracedCMP()


def run_reference_cmp(sequence, D, **kw):
    ref = load_reference()
    cmp = traced_cmp(ref)
    coefficients, residual = cmp.computeCoefficients(sequence, D, **kw)
    t = np.array([a[0] for a in cmp.trace], dtype=np.int32)
    k = np.array([a[1] for a in cmp.trace], dtype=np.int32)
    c = np.array([a[2] for a in cmp.trace])
    return coefficients, residual, dict(t=t, k=k, c=c)
