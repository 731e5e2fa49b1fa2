#!/usr/bin/env python
"""Does the fp16 pre-filter hold on REAL frame-loop banks (temporally coherent memory frames: near-duplicate keys)?
Runs the 1080p / 10k-bank clip of bench.py and prints, per memory read, the fall-back flag and candidate statistics."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tracking-anything-with-deva_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from deva.hip import lib, ops  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
net, sd = bench.build_network(dev)
from workload import synth  # noqa: E402
from deva.inference.inference_core import DEVAInferenceCore  # noqa: E402

size = tuple(int(v) for v in os.environ.get('SIZE', '1080x1920').split('x'))
frames = bench.make_clip(size[0], size[1], int(os.environ.get('FRAMES', 24)), seed=7, device=dev)
cfg = synth.base_config()
core = bench.start_clip(net, cfg, frames, int(os.environ.get('OBJECTS', 1)), dev, lt_prefill=int(os.environ.get('PREFILL', 10000)))
L = lib()
real = ops.affinity_topk
out = (ctypes.c_int64 * 5)()


def tapped(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix=None, splits=None):
    res = real(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix, splits)
    n, hw = n_long + n_work, qk.shape[1]
    if L.deva_affinity_prefilter_enabled(n, hw, k):
        ws = ops._AFF_WS[(qk.device, torch.cuda.current_stream(qk.device).cuda_stream)]
        L.deva_affinity_read_stats(ws.data_ptr(), n, hw, k, out, torch.cuda.current_stream().cuda_stream)
        print(f'  read N={n} HW={hw}: flag {out[0]} largest sub-list {out[1]} (cap 32) largest query {out[2]} mean '
              f'candidates/query {out[3] / 1000:.1f} ranges {out[4]}')
    return res


ops.affinity_topk = tapped
for t in range(1, len(frames)):
    print('frame', t)
    core.step(frames[t])
