"""Per-kernel register / LDS / spill summary from `hipcc -Rpass-analysis=kernel-resource-usage` output.
    hipcc ... --cuda-device-only -c -o /tmp/x.o hscmp_api.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/res.txt
    python tools/kernel_resources.py /tmp/res.txt [name filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filters = sys.argv[2:]
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip()
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if filters and not all(f in dem for f in filters):
        continue

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print('%-150s VGPR %s AGPR %s spill %s scratch %s SGPR %s occ %s' % (dem[:150], g('VGPRs'), g('AGPRs'), g('VGPRs Spill'),
          g(r'ScratchSize \[bytes/lane\]'), g('SGPRs'), g(r'Occupancy \[waves/SIMD\]')))
