#!/usr/bin/env python
"""Per-(kernel, grid) averages of the FETCH_SIZE / WRITE_SIZE passes of tools/convlab/traffic.sh.

FETCH_SIZE is in KiB at 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM): x 2 x 1024 -> bytes;
WRITE_SIZE x 1024 -> bytes.  These are the L2's fabric-side requests: Infinity-Cache hits are included."""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in sorted(glob.glob(out + '/**/*counter_collection.csv', recursive=True)):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r['Counter_Name'] not in ('FETCH_SIZE', 'WRITE_SIZE'):
                    continue
                name = re.sub(r'^void ', '', r['Kernel_Name']).replace('deva::(anonymous namespace)::', '').replace('deva::', '')
                m = re.match(r'(\w+)<([^>]*)>', name)
                short = (m.group(1) + '<' + m.group(2).replace(' ', '') + '>') if m else re.sub(r'\(.*', '', name)
                if 'conv' not in short and 'splitk' not in short:
                    continue
                c = acc[(short, int(r['Grid_Size']), int(r['Workgroup_Size']))][r['Counter_Name']]
                c[0] += float(r['Counter_Value'])
                c[1] += 1
    print(f'{"kernel":48s} {"grid":>8s} {"wg":>5s} {"n":>4s} {"read MB":>9s} {"write MB":>9s}')
    for key in sorted(acc, key=lambda k: -acc[k]['FETCH_SIZE'][0] / max(acc[k]['FETCH_SIZE'][1], 1)):
        f, w = acc[key]['FETCH_SIZE'], acc[key]['WRITE_SIZE']
        rd = f[0] / max(f[1], 1) * 2048 / 1e6
        wr = w[0] / max(w[1], 1) * 1024 / 1e6
        print(f'{key[0]:48s} {key[1]:8d} {key[2]:5d} {f[1]:4d} {rd:9.2f} {wr:9.2f}')


if __name__ == '__main__':
    main()
