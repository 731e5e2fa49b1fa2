#!/usr/bin/env python
"""Launch a few representative deva_conv2d shapes repeatedly (for rocprofv3 --pmc passes and quick
event timing).  Shapes are the heavy layers of the 480p / 5-object and 1080p / 1-object frames."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tracking-anything-with-deva_amd'))
import torch  # noqa: E402
from deva.hip import ops  # noqa: E402

SHAPES = [  # name, cin, cout, k, batch, H, W
    ('up8_4 3x3 256->256 @120x216 x5', 256, 256, 3, 5, 120, 216),
    ('fuser 3x3 512->512 @30x54 x5', 512, 512, 3, 5, 30, 54),
    ('gru 3x3 1024->1536 @30x54 x5', 1024, 1536, 3, 5, 30, 54),
    ('res 1x1 1024->256 @30x54 x1', 1024, 256, 1, 1, 30, 54),
    ('res 3x3 256->256 @30x54 x1', 256, 256, 3, 1, 30, 54),
    ('up8_4 3x3 256->256 @272x480 x1', 256, 256, 3, 1, 272, 480),
    ('res 3x3 64->64 @272x480 x1', 64, 64, 3, 1, 272, 480),
    ('res 1x1 256->64 @272x480 x1', 256, 64, 1, 1, 272, 480),
    ('fuser 3x3 1024->512 @30x54 x5', 1024, 512, 3, 5, 30, 54),
    ('up16_8 3x3 512->256 @60x108 x5', 512, 256, 3, 5, 60, 108),
    ('up16_8 3x3 256->256 @60x108 x5', 256, 256, 3, 5, 60, 108),
    ('res 3x3 128->128 @136x240 x1', 128, 128, 3, 1, 136, 240),
    ('res 3x3 256->256 @68x120 x1', 256, 256, 3, 1, 68, 120),
    ('fuser 3x3 512->512 @68x120 x1', 512, 512, 3, 1, 68, 120),
    ('key 3x3 512->64 @30x54 x1', 512, 64, 3, 1, 30, 54),
    ('shrink 3x3 512->1 @30x54 x1', 512, 1, 3, 1, 30, 54),
    ('pred 3x3 256->1 @120x216 x5', 256, 1, 3, 5, 120, 216),
    ('res 1x1 256->1024 @30x54 x1', 256, 1024, 1, 1, 30, 54),
    ('res 1x1 512->128 @60x108 x1', 512, 128, 1, 1, 60, 108),
    ('res 1x1 128->512 @60x108 x1', 128, 512, 1, 1, 60, 108),
    ('res 1x1 64->256 @120x216 x1', 64, 256, 1, 1, 120, 216),
    ('res 3x3 64->64 @120x216 x1', 64, 64, 3, 1, 120, 216),
    ('res 3x3 128->128 @60x108 x1', 128, 128, 3, 1, 60, 108),
    ('stem 7x7s2 3->64 @480x864 x1', 3, 64, 7, 1, 480, 864),
    ('stem 7x7s2 4->64 @480x864 x5', 4, 64, 7, 5, 480, 864),
]


def main():
    iters = int(os.environ.get('ITERS', 5))
    only = os.environ.get('ONLY')
    dev = torch.device('cuda:0')
    if os.environ.get('NOSPLIT'):  # A/B: no split-K workspace
        tiny = torch.zeros(1, device=dev)
        ops._workspace = lambda device: tiny
    g = torch.Generator().manual_seed(0)
    for name, cin, cout, k, b, h, w in SHAPES:
        if only and only not in name:
            continue
        pc = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k))**0.5,
                           torch.randn(cout, generator=g) * 0.1, device=dev)
        x = ops._alloc((b, cin, h, w), dev)  # guard-banded like every tensor the package allocates
        x.copy_(torch.randn(b, cin, h, w, generator=g))
        if os.environ.get('UNGUARDED'):
            x = x.clone()
        stride = 2 if name.startswith('stem') else 1
        out = ops.conv2d(pc, x, pad=k // 2, stride=stride)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            ops.conv2d(pc, x, pad=k // 2, stride=stride, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        fl = 2.0 * cin * cout * k * k * b * (h // stride) * (w // stride)
        print(f'{name:40s} {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / 1e9:.1f} GF)')


if __name__ == '__main__':
    main()
