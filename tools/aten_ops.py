#!/usr/bin/env python
"""Which ATen / runtime kernels does the steady-state frame loop still launch, and from which line?
    python tools/aten_ops.py [480p|8seg]
torch.profiler over a few frames of the headline loop (bench.py's clip), CPU-side op records with Python stacks: every
aten:: op that is not a view / empty, with its call site inside the package, counted per frame.  """
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tracking-anything-with-deva_amd')]
import torch  # noqa: E402

import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
net, _ = bench.build_network(dev)
from workload import synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else '480p'
frames_n = 12
if which == '480p':
    cfg = synth.base_config(enable_long_term=False, enable_long_term_count_usage=False)
    frames = bench.make_clip(480, 854, 8 + frames_n, seed=100, device=dev)
    core = bench.start_clip(net, cfg, frames, 5, dev)
else:
    cfg = synth.base_config()
    frames = bench.make_clip(1080, 1920, 8 + frames_n, seed=7, device=dev)
    core = bench.start_clip(net, cfg, frames, 3, dev, lt_prefill=10000)
for f in frames[1:8]:
    core.step(f)
torch.cuda.synchronize()
VIEWS = {'aten::empty', 'aten::empty_strided', 'aten::view', 'aten::reshape', 'aten::unsqueeze', 'aten::select', 'aten::slice',
         'aten::as_strided', 'aten::movedim', 'aten::permute', 'aten::unbind', 'aten::_unsafe_view', 'aten::squeeze',
         'aten::expand', 'aten::t', 'aten::transpose', 'aten::flatten', 'aten::empty_like', 'aten::alias', 'aten::detach',
         'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::lift_fresh', 'aten::resolve_conj', 'aten::item',
         'aten::_local_scalar_dense', 'aten::is_nonzero', 'aten::narrow', 'aten::unflatten', 'aten::result_type'}
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            with_stack=True, record_shapes=True) as prof:
    for f in frames[8:]:
        core.step(f)
    torch.cuda.synchronize()
count = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.name in VIEWS:
        continue
    if getattr(ev, 'device_time_total', getattr(ev, 'cuda_time_total', 0)) <= 0:
        continue
    site = next((s for s in (ev.stack or []) if 'tracking-anything-with-deva_amd' in s or 'bench.py' in s), '?')
    site = site.replace(ROOT + '/', '')
    count[(ev.name, site, str(ev.input_shapes)[:80])] += 1
print(f'{which}: ATen ops that launched device work, per {frames_n} frames (memory frame every 5th):')
for (name, site, shapes), n in sorted(count.items(), key=lambda kv: -kv[1]):
    print(f'{n / frames_n:6.2f} /frame  {name:28s} {site}  {shapes}')
