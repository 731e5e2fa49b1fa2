#!/usr/bin/env python
"""debug aid: which queries differ between the two paths of deva_affinity_read, and how"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tracking-anything-with-deva_amd'))
import torch
from deva.hip import lib
dev = torch.device('cuda:0'); L = lib(); st = torch.cuda.current_stream().cuda_stream
for n, hw in [(10000, 8160), (6000, 4099), (10000, 4096)]:
    k = 30
    g = torch.Generator().manual_seed(n + hw)
    key = torch.randn(n, 64, generator=g).to(dev); shr = (torch.rand(n, generator=g) + 1).to(dev)
    qk, qe = torch.randn(64, hw, generator=g).to(dev), torch.rand(64, hw, generator=g).to(dev)
    scratch = torch.empty((L.deva_affinity_read_scratch(n, hw, k),), dtype=torch.int64, device=dev)
    res = {}
    for mode in (0, 1, 1):
        L.deva_affinity_force_prefilter(mode)
        idx = torch.empty((hw, k), dtype=torch.int32, device=dev); w = torch.empty((hw, k), dtype=torch.float32, device=dev)
        L.deva_affinity_read(None, None, 0, key.data_ptr(), shr.data_ptr(), n, qk.data_ptr(), qe.data_ptr(), hw, k,
                             scratch.data_ptr(), idx.data_ptr(), w.data_ptr(), None, None, None, 0, st)
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((idx.cpu(), w.cpu()))
    (i0, w0), (i1, w1), (i2, w2) = res[0][0], res[1][0], res[1][1]
    bad = torch.nonzero((i0 != i1).any(1)).flatten().tolist()
    print(f'N={n} HW={hw}: differing queries {bad}; pre-filter run-to-run identical {bool(torch.equal(i1, i2))}')
    for q in bad[:3]:
        print('  fp32 :', i0[q].tolist()); print('  pre  :', i1[q].tolist())
        print('  w fp32', [f'{v:.4f}' for v in w0[q].tolist()][:8], 'w pre', [f'{v:.4f}' for v in w1[q].tolist()][:8])
        a, b = set(i0[q].tolist()), set(i1[q].tolist())
        print('  only fp32', sorted(a - b), 'only pre', sorted(b - a))
