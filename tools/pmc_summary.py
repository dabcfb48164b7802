#!/usr/bin/env python
"""rocprofv3 --pmc passes (tools/profile_pmc.sh) -> the per-kernel summary bench.py / bench_hsc.py read.

usage: pmc_summary.py --source '<what was profiled>' <pass-dir> [<pass-dir> ...] > profiles/pmc_summary.json

Per kernel (averages over its launches): every counter found, the launch duration seen by the profiler, and
  read_bytes  = 2 x FETCH_SIZE x 1024   (FETCH_SIZE is in KiB; on gfx950 it reports half of the bytes of a streaming read --
                                         MI355X_MICROARCH.md "HBM"; re-checked on prepare_kernel, which reads its batch once.
                                         The x2 is calibrated for wide streaming reads: prepare, corr_init, the dense scans of the
                                         loop prologues.  For the gathers of the level >= 1 kernels -- 4 to 32 bytes per
                                         request -- it is an UPPER bound: the counter may already hold the full sectors.)
  write_bytes = WRITE_SIZE x 1024
  hbm_bytes_per_launch = read_bytes + write_bytes
  mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
FETCH_SIZE and WRITE_SIZE come from separate passes, never combined with a trace domain other than --kernel-trace.
`_csrc_sha256` (tools/csrc_digest.py) stamps the summary with the sources it was collected on: the benches drop the traffic figures
of a summary whose stamp differs from the tree they run from."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_digest import csrc_digest  # noqa: E402


def short(name):
    base = name.split('(')[0].replace('void ', '').replace('hscmp::', '')
    m = re.match(r'(\w+)', base)
    key = m.group(1) if m else base
    pol = re.search(r'(Mfma|Sparse|Generic|DictList|Dense)Recorr', name)
    if key == 'iterate_kernel' and pol:
        key += '[' + pol.group(1).lower() + (',x4' if re.search(r'Recorr<.*, 4>', name) else '') + ']'
    rp = re.search(r'Rp(Mfma|Sparse)', name)
    if key == 'iterate_rp_kernel' and rp:
        key += '[' + rp.group(1).lower() + ']'
    return key


def main():
    args = sys.argv[1:]
    source = None
    if args and args[0] == '--source':
        source, args = args[1], args[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in args:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
                if r['Dispatch_Id'] not in seen:
                    seen.add(r['Dispatch_Id'])
                    dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    out = {'_source': source, '_csrc_sha256': csrc_digest()}
    for k, cs in agg.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d['launches_seen'] = len(dur[k])
        d['ms_under_profiler'] = sum(dur[k]) / len(dur[k]) / 1e6
        if 'FETCH_SIZE' in d:
            d['read_bytes'] = 2.0 * 1024.0 * d['FETCH_SIZE']
        if 'WRITE_SIZE' in d:
            d['write_bytes'] = 1024.0 * d['WRITE_SIZE']
        if 'read_bytes' in d and 'write_bytes' in d:
            d['hbm_bytes_per_launch'] = d['read_bytes'] + d['write_bytes']
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d:
            d['mfma_busy_fraction'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * d['GRBM_GUI_ACTIVE'] / 8.0)
        out[k] = d
    # names bench.py looks up
    for k in list(out):
        if k.startswith('iterate_kernel[mfma') or k.startswith('iterate_rp_kernel[mfma'):
            out.setdefault('iterate_kernel', out[k])
            if 'hbm_bytes_per_launch' in out[k]:
                out['level0_loop_hbm_bytes_per_launch'] = out[k]['hbm_bytes_per_launch']
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
