#!/usr/bin/env python
"""Event-timed pointwise ops at the shapes of the frame (tuning aid):  python tools/time_ops.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tracking-anything-with-deva_amd')]
import torch  # noqa: E402

from deva.hip import lib, ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def t(*shape):
    x = ops._alloc(shape, dev)
    x.copy_(torch.randn(*shape, generator=g))
    return x


def timeit(name, fn, iters=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f'{name:60s} {a.elapsed_time(b) / iters * 1e3:8.1f} us')


L = lib()
st = torch.cuda.current_stream().cuda_stream
for tag, (no, h, w) in (('480p/5', (5, 30, 54)), ('1080p/11', (11, 68, 120))):
    x = t(no, 512, h, w)
    w1, b1 = torch.randn(32, 512, generator=g).to(dev) * 0.05, torch.randn(32, generator=g).to(dev) * 0.1
    w2, b2 = torch.randn(512, 32, generator=g).to(dev) * 0.2, torch.randn(512, generator=g).to(dev) * 0.1
    sp = ops.pack_conv(torch.randn(1, 2, 7, 7, generator=g) * 0.2, torch.randn(1, generator=g) * 0.1, None, dev)
    timeit(f'cbam (5 launches) {tag}', lambda: ops.cbam(x, w1, b1, w2, b2, sp))
    avg, mx, sc = (torch.empty(no, 512, device=dev) for _ in range(3))
    timeit(f'  global_avgmax {tag}', lambda: L.deva_global_avgmax(x.data_ptr(), avg.data_ptr(), mx.data_ptr(), no * 512, h * w, st))
    timeit(f'  cbam_mlp {tag}', lambda: L.deva_cbam_mlp(avg.data_ptr(), mx.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                        b2.data_ptr(), sc.data_ptr(), no, 512, 32, st))
    pooled = t(no, 2, h, w)
    timeit(f'  cbam_channel_pool {tag}', lambda: L.deva_cbam_channel_pool(x.data_ptr(), sc.data_ptr(), pooled.data_ptr(), no, 512, h * w, st))
    gate = t(no, 1, h, w)
    out = t(no, 512, h, w)
    timeit(f'  cbam_apply {tag}', lambda: L.deva_cbam_apply(x.data_ptr(), sc.data_ptr(), gate.data_ptr(), out.data_ptr(), no, 512, h * w, st))
    vals, hh = t(no, 1536, h, w), t(no, 512, h, w)
    timeit(f'gru_update {tag}', lambda: ops.gru_update(vals, hh))
    p8 = t(no, 256, 2 * h, 2 * w)
    d4 = t(1, 256, 4 * h, 4 * w)
    timeit(f'upsample2x_add p8->p4 {tag}', lambda: ops.upsample2x_add(p8, d4))
    timeit(f'upsample2x_add_ds2 p8->p4 {tag}', lambda: ops.upsample2x_add_ds2(p8, d4))
    timeit(f'area_downsample(p8, 2) {tag}', lambda: ops.area_downsample(p8, 2))
    p4 = t(no, 256, 4 * h, 4 * w)
    timeit(f'area_downsample(p4, 4) {tag}', lambda: ops.area_downsample(p4, 4))
    lm = t(no, 16 * h, 16 * w)
    timeit(f'area_downsample(last_mask, 16) {tag}', lambda: ops.area_downsample(lm, 16))
    del x, out, vals, hh, p8, d4, p4, lm
    torch.cuda.empty_cache()
