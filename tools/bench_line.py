#!/usr/bin/env python
"""One `also` line of bench.py on its own, with the per-layer convolution table of its event-timed replay:
    python tools/bench_line.py 8seg|1080p1|480p5 [fp32|split|split_all|amp] [out.json]
(tuning aid: the same functions bench.py runs, nothing else in the process)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else '8seg'
    prec = sys.argv[2] if len(sys.argv) > 2 else 'split'
    out = sys.argv[3] if len(sys.argv) > 3 else None
    torch.set_grad_enabled(False)
    device = torch.device('cuda:0')
    from workload import synth
    from deva.model.network import DEVA
    net, sd = bench.build_network(device, amp=prec == 'amp', split=prec.startswith('split'),
                                  **({'split_key_encoder': True} if prec == 'split_all' else {}))
    layers = {}
    real_report = bench.conv_roofline_report

    def report(ct, frames):
        layers['rows'] = ct.per_layer(frames)
        return real_report(ct, frames)

    bench.conv_roofline_report = report
    steps, warmup = int(os.environ.get('DEVA_LINE_STEPS', 25)), int(os.environ.get('DEVA_LINE_WARMUP', 6))  # (PMC passes: fewer)
    if what == '8seg':
        fps, state = bench.run_1080p_segments(net, device, steps=steps, warmup=warmup, segments=8, conv_roofline=True)
    elif what == '1080p1':
        fps, state = bench.run_1080p(net, device, steps=25, warmup=6, detections=False, conv_roofline=True)
    else:
        class A:
            height, width, objects, steps, warmup = 480, 854, 5, 40, 5
        cfg = synth.base_config(enable_long_term=False, enable_long_term_count_usage=False)
        fps, state = bench.run_480p_headline(net, device, cfg, A)
    print(f'{what} {prec}: {fps:.2f} FPS ({1e3 / fps:.2f} ms/frame)')
    print(json.dumps(state.get('conv_roofline'), indent=1))
    rows = layers.get('rows', [])
    tot = sum(r['ms_per_frame'] for r in rows)
    print(f'conv ms per frame {tot:.2f}')
    for r in rows[:40]:
        print('  %4d>%4d k%d s%d b%2d %3dx%3d  calls %.2f  %7.3f ms  %6.1f TF' % (r['cin'], r['cout'], r['k'], r['stride'], r['batch'], r['oh'], r['ow'],
                                                                                r['calls_per_frame'], r['ms_per_frame'], r['tflops']))
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, 'w') as f:
            json.dump({'fps': fps, 'state': state, 'layers': rows}, f, indent=1, default=str)


if __name__ == '__main__':
    main()
