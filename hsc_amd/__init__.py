"""Import alias: `hsc_amd` -> the package directory `hierarchical-sparse-coding_amd/` (whose name
is not a valid Python identifier).  `import hsc_amd.modeling` is the drop-in for the reference's
`hsc.modeling` on the matching-pursuit hot path."""
import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'hierarchical-sparse-coding_amd')
if not _os.path.isdir(_impl):
    raise ImportError('hsc_amd: implementation directory not found: %s' % _impl)
__path__.insert(0, _impl)
del _os
