#!/usr/bin/env python
"""Event-timed launches of the fused affinity kernel at the microbench shapes of SURVEY.md §8d."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tracking-anything-with-deva_amd'))
import torch  # noqa: E402
from deva import hip  # noqa: E402
from deva.hip import lib, ops  # noqa: E402

if os.environ.get('DEVA_HIP_LIB'):  # e.g. the probe build tools/probe/libdeva_hip_probes.so
    hip.LIB_PATH = os.path.abspath(os.environ['DEVA_HIP_LIB'])

SHAPES = [(10000, 1620), (24580, 1620), (10000, 8160), (83440, 8160)]


def main():
    iters = int(os.environ.get('ITERS', 5))
    only = os.environ.get('ONLY')
    dev = torch.device('cuda:0')
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    shapes = SHAPES
    if os.environ.get('SHAPES'):  # e.g. SHAPES=1620x1620,6480x1620
        shapes = [tuple(int(v) for v in t.split('x')) for t in os.environ['SHAPES'].split(',')]
    for n, hw in shapes:
        if only and only != f'{n}x{hw}':
            continue
        key = torch.randn(n, 64, generator=g).to(dev)
        shr = (torch.rand(n, generator=g) + 1).to(dev)
        qk, qe = torch.randn(64, hw, generator=g).to(dev), torch.rand(64, hw, generator=g).to(dev)
        k = 30
        splits = int(os.environ.get('SPLITS', L.deva_affinity_default_splits(n, hw)))
        part = torch.empty((L.deva_affinity_workspace(hw, k, splits),), dtype=torch.int64, device=dev)
        idx = torch.empty((hw, k), dtype=torch.int32, device=dev)
        w = torch.empty((hw, k), dtype=torch.float32, device=dev)

        def run(fin=True):
            L.deva_affinity_topk(None, None, 0, key.data_ptr(), shr.data_ptr(), n, qk.data_ptr(), qe.data_ptr(), hw,
                                 k, splits, part.data_ptr(), st)
            if fin:
                L.deva_affinity_finalize(part.data_ptr(), hw, k, splits, idx.data_ptr(), w.data_ptr(), None, st)

        fin_ok = not os.environ.get('DEVA_AFFINITY_ABLATE')
        run(fin_ok)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            run(fin_ok)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        s.record()
        for _ in range(iters):
            run(False)
        e.record()
        torch.cuda.synchronize()
        ms_topk = s.elapsed_time(e) / iters
        fl = 4.0 * 64 * n * hw
        print(f'affinity N={n:6d} HW={hw:5d} splits={splits:2d}: {ms * 1e3:9.1f} us (filter {ms_topk * 1e3:7.1f} us)  '
              f'{fl / ms / 1e9:6.1f} TFLOP/s ({fl / 1e9:.1f} GF)')


if __name__ == '__main__':
    main()
