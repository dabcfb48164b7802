"""Loads tools/csrc_digest.py by path (tools/ is not a package and must not be put in front of sys.path: it holds scripts whose names
shadow root modules, e.g. tools/bench_hsc.py)."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location('_csrc_digest', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'csrc_digest.py'))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
csrc_digest = _mod.csrc_digest
load_pmc_summary = _mod.load_pmc_summary
