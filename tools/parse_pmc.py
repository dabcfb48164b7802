#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs into profiles/pmc_summary.json.

HBM traffic per launch follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB, collected
in SEPARATE passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming
read (calibrated here on prepare_kernel, which reads the 268 MB batch exactly once: raw FETCH_SIZE
131 MB) => read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact.

usage: parse_pmc.py <fetch.csv> <write.csv> [<mfma.csv>] > profiles/pmc_summary.json
"""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def short(name):
    for key in ('corr_init_mfma_kernel', 'iterate_kernel', 'prepare_kernel', 'corr_init_generic_kernel'):
        if key in name:
            return key
    return None


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    mfma = load(sys.argv[3]) if len(sys.argv) > 3 else {}
    out = {}
    for (kn, cn), v in fetch.items():
        s = short(kn)
        if s and cn == 'FETCH_SIZE':
            out.setdefault(s, {})['fetch_size_kib_raw'] = v
            out[s]['read_bytes'] = 2.0 * v * 1024.0
    for (kn, cn), v in write.items():
        s = short(kn)
        if s and cn == 'WRITE_SIZE':
            out.setdefault(s, {})['write_size_kib'] = v
            out[s]['write_bytes'] = v * 1024.0
    for s in out:
        out[s]['hbm_bytes_per_launch'] = out[s].get('read_bytes', 0.0) + out[s].get('write_bytes', 0.0)
    for (kn, cn), v in mfma.items():
        s = short(kn)
        if s:
            out.setdefault(s, {})[cn] = v
    for s, d in out.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            d['mfma_busy_fraction'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * d['GRBM_GUI_ACTIVE'] / 8.0)
        if 'SQ_INSTS_VALU_MFMA_MOPS_F32' in d:
            d['mfma_flop'] = 512.0 * d['SQ_INSTS_VALU_MFMA_MOPS_F32']
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
