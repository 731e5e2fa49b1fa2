#!/usr/bin/env python
"""Where a tile's time goes in the workgroup-shared affinity kernel: cycle stamps (s_memtime) of the first eight
workgroups' waves at the phase boundaries of their first 64 tiles.  Needs a probe build of the library:
    hipcc ... -DDEVA_AFFINITY_PROBES -c affinity.hip ; link as tools/probe/libdeva_hip_probes.so  (see README.md)
Phases: 0 loop top -> 1 barrier A + list lengths read -> 2 prune (if any) -> 3 MFMAs done -> 4 prefetch issued,
shrinkage read -> 5 barrier C passed -> 6 scores + slot reservations -> 7 entries written."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tracking-anything-with-deva_amd'))
import torch  # noqa: E402
from deva import hip  # noqa: E402

hip.LIB_PATH = os.path.join(ROOT, 'tools', 'probe', 'libdeva_hip_probes.so')
NAMES = ['barrier A + lengths', 'prune', 'MFMAs (+ms store)', 'prefetch + ms read', 'barrier C', 'score + reserve',
         'write entries', '(loop back)']


def main():
    n, hw = (int(v) for v in os.environ.get('SHAPE', '10000x8160').split('x'))
    shape = int(os.environ.get('DEVA_AFFINITY_SHAPE', 4))
    nw = 8 if shape == 8 else 4
    dev = torch.device('cuda:0')
    L = hip.lib()
    L.deva_affinity_set_probe.argtypes = [ctypes.c_void_p]
    L.deva_affinity_force_shape(shape)
    g = torch.Generator().manual_seed(0)
    key = torch.randn(n, 64, generator=g).to(dev)
    shr = (torch.rand(n, generator=g) + 1).to(dev)
    qk, qe = torch.randn(64, hw, generator=g).to(dev), torch.rand(64, hw, generator=g).to(dev)
    k = 30
    splits = L.deva_affinity_default_splits(n, hw)
    part = torch.empty((L.deva_affinity_workspace(hw, k, splits),), dtype=torch.int64, device=dev)
    probe = torch.zeros((8, nw, 64, 8), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = L.deva_affinity_topk(None, None, 0, key.data_ptr(), shr.data_ptr(), n, qk.data_ptr(), qe.data_ptr(), hw,
                                  k, splits, part.data_ptr(), st)
        assert rc == 0, L.deva_hip_last_error()

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.deva_affinity_set_probe(probe.data_ptr())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    run()
    e.record()
    torch.cuda.synchronize()
    L.deva_affinity_set_probe(None)
    us = s.elapsed_time(e) * 1e3
    t = probe.cpu().double()  # [block][wave][tile][stamp]
    n_it = int((t[..., 3] > 0).all(dim=1).all(dim=0).sum())  # tiles every stamped wave worked on
    t = t[:, :, :n_it]
    span = (t[:, :, -1, 7] - t[:, :, 0, 0]).mean().item()
    print(f'shape {shape}, N={n} HW={hw} splits={splits}: kernel {us:.1f} us (events, one launch); {n_it} tiles per wave; '
          f'stamped span {span:.0f} ticks per wave = {span / n_it:.1f} ticks per tile')
    d = torch.zeros(8)
    d[:7] = (t[..., 1:] - t[..., :7]).mean(dim=(0, 1, 2))
    d[7] = (t[:, :, 1:, 0] - t[:, :, :-1, 7]).mean()
    tot = d.sum().item()
    for i, name in enumerate(NAMES):
        print(f'  {i} -> {(i + 1) % 8}  {name:22s} {d[i].item():8.1f} ticks  {100 * d[i].item() / tot:5.1f} %')
    pr = (t[..., 2] - t[..., 1])
    print(f'  tiles with a prune round: {int((pr > pr.median() * 4 + 20).sum())} of {pr.numel()}')
    per_tile = (t[:, :, :, 7] - t[:, :, :, 0])
    print('  per-tile ticks, wave 0 of workgroup 0:', ' '.join(f'{v:.0f}' for v in per_tile[0, 0, :n_it].tolist()))
    print(f'  ticks per us (span of wave 0 / kernel us, upper bound): {span / us:.1f}')


if __name__ == '__main__':
    main()
