#!/usr/bin/env python
"""Host-side profile of the 480p / 1-object frame loop (the launch-gap regime): wall time per frame, GPU kernel time per
frame (events around the whole loop vs the sum of kernel durations is rocprof's job), and a cProfile of the Python side of
`DEVAInferenceCore.step`.

    python tools/host_profile_480p1.py [frames]      (needs a GPU)
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tracking-anything-with-deva_amd'))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    device = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    net, _ = bench.build_network(device)
    from workload import synth
    cfg = synth.base_config()
    frames = bench.make_clip(480, 854, 1 + 10 + 2 * n, seed=100, device=device)
    core = bench.start_clip(net, cfg, frames, 1, device)
    for t in range(1, 11):
        core.step(frames[t])
    torch.cuda.synchronize()
    # 1) free-running loop: wall per frame, and how far the host runs ahead of the GPU
    t0 = time.perf_counter()
    for t in range(11, 11 + n):
        core.step(frames[t])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'free-running: host issued {n} frames in {t_host * 1e3 / n:.3f} ms/frame; GPU done after {t_all * 1e3 / n:.3f} ms/frame '
          f'-> {n / t_all:.1f} FPS; host share {t_host / t_all:.2f} (close to 1.0 = the Python side is the bottleneck)')
    # 2) cProfile of the host side
    pr = cProfile.Profile()
    pr.enable()
    for t in range(11 + n, 11 + 2 * n):
        core.step(frames[t])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(28)
    st.sort_stats('tottime').print_stats(22)


if __name__ == '__main__':
    main()
