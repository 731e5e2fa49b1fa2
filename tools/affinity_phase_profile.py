#!/usr/bin/env python
"""Per-phase cycle totals of affinity_topk_kernel (needs the -DDEVA_AFF_PROFILE build of affinity.hip:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DDEVA_AFF_PROFILE -shared \
        tracking-anything-with-deva_amd/csrc/{affinity,runtime}.hip -o tools/_prof/libdeva_affprof.so)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, 'tools', '_prof', 'libdeva_affprof.so'))
L.deva_affinity_workspace.restype = ctypes.c_int64
vp, ci = ctypes.c_void_p, ctypes.c_int
L.deva_affinity_topk.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, vp, vp]
L.deva_aff_prof_read.argtypes = [vp, ci]

dev = torch.device('cuda:0')
for n, hw, splits in [(10000, 8160, 4), (10000, 1620, 16), (83440, 8160, 4)]:
    g = torch.Generator().manual_seed(0)
    key = torch.randn(n, 64, generator=g).to(dev)
    shr = (torch.rand(n, generator=g) + 1).to(dev)
    qk, qe = torch.randn(64, hw, generator=g).to(dev), torch.rand(64, hw, generator=g).to(dev)
    k = 30
    part = torch.empty((L.deva_affinity_workspace(hw, k, splits),), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        rc = L.deva_affinity_topk(None, None, 0, key.data_ptr(), shr.data_ptr(), n, qk.data_ptr(), qe.data_ptr(),
                                  hw, k, splits, part.data_ptr(), st)
        assert rc == 0
    torch.cuda.synchronize()
    waves = ((hw + 127) // 128) * splits * 4
    host = np.zeros((waves, 4), dtype=np.int64)
    L.deva_aff_prof_read(host.ctypes.data, waves)
    live = host[host.sum(1) > 0]
    tiles = ((n + 31) // 32 + splits - 1) // splits
    m = live.mean(0)
    print(f'N={n} HW={hw} splits={splits}: tiles/wave={tiles} waves={len(live)}  cycles per tile: '
          f'prune+tau {m[0] / tiles:7.0f}  operand+prefetch {m[1] / tiles:7.0f}  mfma {m[2] / tiles:7.0f}  '
          f'epilogue+append {m[3] / tiles:7.0f}  total/tile {m.sum() / tiles:7.0f}')
