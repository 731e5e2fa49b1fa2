#!/usr/bin/env python
"""Best (tile, split target) per layer from the CSVs of tools/convlab/sweep.sh."""
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
res = defaultdict(dict)
calls, gfs = {}, {}
order = []
for f in sorted(glob.glob(os.path.join(d, '*.csv'))):
    cfg = os.path.basename(f)[:-4]
    for line in open(f):
        _, name, li, us, gf, c = line.rstrip('\n').split(',')
        if name not in calls:
            order.append(name)
        res[name][cfg] = float(us)
        calls[name], gfs[name] = float(c), float(gf)
base = 't0_s-1'
tot_base = tot_best = 0.0
for name in order:
    r = res[name]
    best = min(r, key=r.get)
    b = r.get(base, float('nan'))
    tot_base += b * calls[name]
    tot_best += r[best] * calls[name]
    top = sorted(r.items(), key=lambda kv: kv[1])[:4]
    print(f'{name:34s} default {b:8.1f} us | best {best:12s} {r[best]:8.1f} us ({gfs[name] / r[best] * 1e3:6.1f} TF) | ' +
          ' '.join(f'{k}:{v:.1f}' for k, v in top))
print(f'frame: default policy {tot_base:.1f} us, best-per-layer {tot_best:.1f} us')
