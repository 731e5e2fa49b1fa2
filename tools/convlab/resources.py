#!/usr/bin/env python
"""Kernel-resource table of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/convlab/resources.py tracking-anything-with-deva_amd/csrc/conv_mfma.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(root, 'include'), '-c',
       '--cuda-device-only', sys.argv[1], '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.split('\n'):
    m = re.search(r'remark:\s*([A-Za-z ]+(?:\[[^\]]*\])?): (.*?) \[-Rpass', l)
    if not m:
        if 'error' in l:
            print(l)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = {'name': v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    n = re.sub(r'^_ZN4deva12_GLOBAL__N_1\d+', '', r['name'])
    n = re.sub(r'EEvNS.*', '', n).replace('ELi', ',').replace('ILi', '<')
    print('%-46s vgpr %4s agpr %3s spill %3s sgpr %4s occ %2s lds %s' % (
        n, r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('SGPRs'), r.get('Occupancy [waves/SIMD]'),
        r.get('LDS Size [bytes/block]')))
