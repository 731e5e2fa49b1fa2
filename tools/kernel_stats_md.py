#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV of a `bench.py ... --no_extra --no_affinity` run (one workload: the 480p / 5-object
headline loop and its replays) -> per-kernel table in markdown + convolution time per frame.

    python tools/kernel_stats_md.py gpurun_out/r04/trace  "command line"  ["workload label" [bench.json of the traced run]]  > profiles/r04/kernel_stats.md
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void ', '', name).replace('deva::(anonymous namespace)::', '').replace('deva::', '')
    name = re.sub(r'\(.*$', '', name)
    return name[:100]


def main():
    d, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    label = sys.argv[3] if len(sys.argv) > 3 else '480p / 5-object headline loop'
    gflop = 1000.54  # per frame at 480p / 5 objects (bench.py roofline.gflop_per_frame)
    if len(sys.argv) > 4:  # the bench line of the traced run: take the figure from there
        import json
        with open(sys.argv[4]) as f:
            gflop = json.loads(f.read().strip().split('\n')[-1])['roofline']['gflop_per_frame']
    rows = []
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    probe = [0, 0.0]
    for r in rows:
        n = short(r['Kernel_Name'])
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if n.startswith('mfma_probe'):  # bench.py's matrix-pipe probe (deva_probe_mfma_f32): not part of the frame loop
            probe[0] += 1
            probe[1] += dur
            continue
        a = agg[n]
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    frames = agg.get('upsample4x_softmax_kernel', [0])[0] + 1  # one decoder pass per propagated frame + the annotated frame
    is_conv = lambda n: n.startswith(('conv_', 'splitk_reduce'))
    conv = sum(a[1] for n, a in agg.items() if is_conv(n))
    mfma = sum(a[1] for n, a in agg.items() if n.startswith(('conv_mfma', 'conv_igemm', 'conv_wino')))
    wino = sum(a[1] for n, a in agg.items() if n.startswith('conv_wino'))
    aff = sum(a[1] for n, a in agg.items() if n.startswith('affinity') or n.startswith('readout'))
    print(f'# rocprofv3 --kernel-trace of `{cmd}`\n')
    print(f'One workload in the trace: the {label} ({frames} frames: annotated + warm-up + timed + event-timed '
          f'replay + the per-frame-synchronised replay).  {len(rows)} dispatches, {total / 1e3:.1f} ms of kernel time'
          + (f' (+ {probe[0]} launches = {probe[1] / 1e3:.1f} ms of the matrix-pipe probe, left out of every figure below)' if probe[0] else '') + '.\n')
    print(f'* convolution kernels (conv_wino / conv_mfma / conv_igemm + splitk_reduce + conv_cout1 / conv3x3_cout1_rows): {conv / 1e3:.1f} ms = '
          f'**{conv / 1e3 / frames:.3f} ms per frame** = {100 * conv / total:.1f} % of GPU time '
          f'(MFMA kernels alone {mfma / 1e3 / frames:.3f} ms per frame, of which the Winograd kernels {wino / 1e3 / frames:.3f})')
    print(f'* memory read (affinity_* + readout_sparse): {aff / 1e3:.2f} ms = {aff / 1e3 / frames:.3f} ms per frame = {100 * aff / total:.2f} %')
    print(f'* {gflop:.2f} GF of convolution per frame (bench.py roofline.gflop_per_frame) / {conv / 1e3 / frames:.3f} ms = '
          f'{gflop / (conv / 1e3 / frames):.1f} TFLOP/s = {gflop / (conv / 1e3 / frames) / 157.3:.3f} of the fp32-MFMA peak, kernel time '
          'only (the event pairs of bench.py also bracket the launch gaps inside a deva_conv2d call)\n')
    print('| kernel | calls | total ms | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|')
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f'| `{n}` | {a[0]} | {a[1] / 1e3:.2f} | {100 * a[1] / total:.1f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} |')


if __name__ == '__main__':
    main()
