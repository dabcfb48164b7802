"""Static instruction counts between HSCMP_MARK comments of one kernel's assembly (analysis build: -DHSCMP_MARKS -S).
usage: count_marks.py <dev.s> <mangled-name-substring>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l and l.rstrip().endswith(sys.argv[2].split()[-1]) or (l.startswith('_Z') and sys.argv[2] in l and ':' in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))


def cls(l):
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        return None
    op = l.split()[0]
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')):
        return 'vmem'
    return 'other'


cur = 'entry'
counts = collections.OrderedDict()
for l in lines[start:end]:
    m = re.search(r'; HSCMP_MARK (\w+)', l)
    if m:
        cur = m.group(1) + '@%d' % (len([k for k in counts if k.startswith(m.group(1) + '@')]))
        continue
    c = cls(l)
    if c:
        counts.setdefault(cur, collections.Counter())[c] += 1
for k, v in counts.items():
    print('%-20s %s' % (k, dict(v)))
