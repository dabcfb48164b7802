#!/usr/bin/env python
"""Per-kernel averages of every counter found under the given rocprofv3 --pmc output directories.
usage: parse_pmc2.py <dir> [<dir> ...]   (prints a table; --json writes a dict)"""
import collections
import csv
import glob
import json
import os
import sys

dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            short = name.split('(')[0].replace('void hscmp::', '')[:70]
            agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[short].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {}
for k in agg:
    out[k] = {c: sum(v) / len(v) for c, v in agg[k].items()}
    out[k]['_ms'] = sum(dur[k]) / len(dur[k]) / 1e6
    out[k]['_launches'] = len(dur[k])
if '--json' in sys.argv:
    print(json.dumps(out, indent=1))
else:
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['_ms']):
        print(k, ' ms %.3f' % v['_ms'])
        for c, x in sorted(v.items()):
            if not c.startswith('_'):
                print('    %-32s %.4g' % (c, x))
