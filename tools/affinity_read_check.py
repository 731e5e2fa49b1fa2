#!/usr/bin/env python
"""deva_affinity_read (fp16 pre-filter + exact fp32 re-scoring) against the fp32 kernels on the same inputs:
bit-identical idx / weight / usage, fall-back flag, and event-timed launches of both paths.
    SHAPES=10000x8160,83440x8160 ITERS=10 python tools/affinity_read_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tracking-anything-with-deva_amd'))
import torch  # noqa: E402
from deva.hip import lib  # noqa: E402

SHAPES = [(2048, 1620), (10000, 1620), (24580, 1620), (10000, 8160), (83440, 8160), (50000, 32400)]


def main():
    iters = int(os.environ.get('ITERS', 10))
    dev = torch.device('cuda:0')
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    shapes = SHAPES
    if os.environ.get('SHAPES'):
        shapes = [tuple(int(v) for v in t.split('x')) for t in os.environ['SHAPES'].split(',')]
    scale = float(os.environ.get('KEY_SCALE', 1.0))
    k = int(os.environ.get('TOPK', 30))
    for n, hw in shapes:
        g = torch.Generator().manual_seed(n + hw)
        key = (torch.randn(n, 64, generator=g) * scale).to(dev)
        shr = (torch.rand(n, generator=g) + 1).to(dev)
        qk, qe = (torch.randn(64, hw, generator=g) * scale).to(dev), torch.rand(64, hw, generator=g).to(dev)
        scratch = torch.empty((L.deva_affinity_read_scratch(n, hw, k),), dtype=torch.int64, device=dev)
        out = {}
        for mode in (0, 1):
            L.deva_affinity_force_prefilter(mode)
            idx = torch.empty((hw, k), dtype=torch.int32, device=dev)
            w = torch.empty((hw, k), dtype=torch.float32, device=dev)
            fix = torch.zeros(n, dtype=torch.int64, device=dev)

            def run(fix_ptr):
                rc = L.deva_affinity_read(None, None, 0, key.data_ptr(), shr.data_ptr(), n, qk.data_ptr(), qe.data_ptr(), hw,
                                          k, scratch.data_ptr(), idx.data_ptr(), w.data_ptr(), fix_ptr, None, None, 0, st)
                assert rc == 0, L.deva_hip_last_error()

            run(fix.data_ptr())
            torch.cuda.synchronize()
            flag = L.deva_affinity_read_flag(scratch.data_ptr(), st) if mode else 0
            for _ in range(2):
                run(None)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                run(None)
            e.record()
            torch.cuda.synchronize()
            out[mode] = (idx.clone(), w.clone(), fix.clone(), s.elapsed_time(e) / iters * 1e3, flag)
        L.deva_affinity_force_prefilter(1)
        (i0, w0, f0, t0, _), (i1, w1, f1, t1, flag) = out[0], out[1]
        same = bool(torch.equal(i0, i1)) and bool(torch.equal(w0.view(torch.int32), w1.view(torch.int32))) and bool(torch.equal(f0, f1))
        bad_q = int((i0 != i1).any(1).sum())
        fl = 4.0 * 64 * n * hw
        print(f'read N={n:6d} HW={hw:5d} k={k}: fp32 kernels {t0:8.1f} us | pre-filter {t1:8.1f} us ({fl / t1 / 1e6:6.1f} TFLOP/s '
              f'fp32-equivalent, {t0 / t1:4.2f}x)  fall-back flag {flag}  bit-identical {same}'
              + ('' if same else f'  QUERIES DIFFERING {bad_q}'))


if __name__ == '__main__':
    main()
