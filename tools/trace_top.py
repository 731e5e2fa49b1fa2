#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSVs of any run of the frame loop -> kernels by total time, per propagated frame
(frames = launches of upsample4x_softmax_kernel: one decoder pass each):  python tools/trace_top.py <dir> [top N] [title]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void ', '', name).replace('deva::(anonymous namespace)::', '').replace('deva::', '')
    m = re.match(r'(\w+)<([^>]*)>', name)
    if m and m.group(1).startswith('conv_'):
        return m.group(1) + '<' + m.group(2).replace(' ', '') + '>'
    return re.sub(r'[<(].*$', '', name)[:80]


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    title = sys.argv[3] if len(sys.argv) > 3 else d
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[short(r['Kernel_Name'])]
            a[0] += 1
            a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    frames = max(agg.get('upsample4x_softmax_kernel', [1])[0], 1)
    total = sum(a[1] for a in agg.values())
    fam = defaultdict(float)
    for n, a in agg.items():
        key = ('conv (f16 pipes)' if n.startswith('conv_f16') else 'conv (fp32 MFMA / VALU heads / split-K reduce)' if n.startswith(('conv', 'splitk'))
               else 'memory read (affinity_*, readout_sparse)' if n.startswith(('affinity', 'readout')) else
               'ATen / runtime (at::native, rocclr fill / copy)' if ('at::' in n or 'rocclr' in n or 'elementwise' in n or 'Functor' in n) else 'other deva kernels')
        fam[key] += a[1]
    print(f'# {title}\n')
    print(f'{frames} propagated frames in the trace, {sum(a[0] for a in agg.values())} dispatches, {total / 1e3:.1f} ms of kernel time = '
          f'{total / 1e3 / frames:.3f} ms per frame.\n')
    print('| family | ms per frame | share |\n|---|---|---|')
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f'| {k} | {v / 1e3 / frames:.3f} | {100 * v / total:.1f} % |')
    print('\n| kernel | launches per frame | us per launch | ms per frame | share |\n|---|---|---|---|---|')
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'| `{n}` | {a[0] / frames:.2f} | {a[1] / a[0]:.1f} | {a[1] / 1e3 / frames:.3f} | {100 * a[1] / total:.1f} % |')


if __name__ == '__main__':
    main()
