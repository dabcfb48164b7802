#!/usr/bin/env python
"""sha256 over the engine's sources (csrc/*.hip, csrc/*.h, include/hscmp.h, the HIPFLAGS / ARCH lines of csrc/Makefile), names and contents in sorted order: the
stamp that ties a PMC summary under profiles/ to the kernels it was collected on.  tools/pmc_summary.py writes it into the summary
(`_csrc_sha256`), bench.py / bench_hsc.py compare it with the tree they run from and report `roofline.traffic = null` +
`pmc_stale = true` when they differ (a kernel change without a PMC re-run must not keep reporting the old bytes)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_digest(root=ROOT):
    csrc = os.path.join(root, 'hierarchical-sparse-coding_amd', 'csrc')
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.h')))
    files.append(os.path.join(root, 'include', 'hscmp.h'))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(fh.read())
        h.update(b'\0')
    # of the Makefile only what the product library is compiled with (its other targets -- the sanitizer build of the host shim --
    # do not change the kernels)
    with open(os.path.join(csrc, 'Makefile')) as fh:
        for line in fh:
            if line.startswith(('HIPFLAGS', 'ARCH')):
                h.update(line.strip().encode() + b'\0')
    return h.hexdigest()


def load_pmc_summary(path, root=ROOT):
    """(summary dict or None, stale flag): stale = the summary carries no stamp or another tree's."""
    import json
    try:
        pmc = json.load(open(path))
    except Exception:
        return None, None
    return pmc, pmc.get('_csrc_sha256') != csrc_digest(root)


if __name__ == '__main__':
    print(csrc_digest())
