#!/usr/bin/env python
"""Long-clip soak on the GPU: 480p, 5 objects, long-term memory on with a small bank so that
consolidations and least-usage evictions happen every few frames; checks that outputs stay finite,
the banks stay within their limits and the allocator does not grow."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tracking-anything-with-deva_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from workload import synth  # noqa: E402


def main():
    frames_n = int(os.environ.get('FRAMES', 400))
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    net, _ = bench.build_network(dev)
    cfg = synth.base_config(mem_every=2, max_long_term_elements=3000, num_prototypes=128)
    stream = synth.FrameStream(480, 854, seed=3)
    frames = [stream.next().to(dev) for _ in range(40)]
    core = bench.start_clip(net, cfg, frames, 5, dev)
    peak0 = None
    for t in range(1, frames_n):
        p = core.step(frames[t % len(frames)])
        if t % 50 == 0:
            torch.cuda.synchronize()
            assert torch.isfinite(p).all(), f'non-finite output at frame {t}'
            mem = core.memory
            lt = {b: mem.long_mem.size(b) for b in mem.long_mem.buckets}
            wk = {b: mem.work_mem.size(b) for b in mem.work_mem.buckets}
            assert all(v <= cfg['max_long_term_elements'] for v in lt.values()), lt
            assert all(v <= cfg['max_mid_term_frames'] * 1620 for v in wk.values()), wk
            alloc = torch.cuda.memory_allocated() / 2**20
            if t == 100:
                peak0 = alloc
            print(f'frame {t}: long {lt} work {wk} allocated {alloc:.0f} MiB argmax classes {p.argmax(0).unique().tolist()}')
            if peak0 is not None:
                assert alloc <= peak0 * 1.2 + 64, 'allocator grows'
    print('soak ok')


if __name__ == '__main__':
    main()
