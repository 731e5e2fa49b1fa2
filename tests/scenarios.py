"""End-to-end propagation scenarios shared by the golden generator (reference), the CPU tests
(oracle) and the GPU tests (HIP runtime).  A scenario drives any object with the
`DEVAInferenceCore.step(image, mask, objects, end=...)` call pattern of
evaluation/eval_vos.py:167."""
import os
import sys
from typing import Callable, Dict, List

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# name -> scenario.  `second` = (frame index, object id) of a later partial annotation that
# introduces a new object (creates a second memory bucket, kv_memory_store.py:82-89).
E2E: Dict[str, Dict] = {
    # long-term on; 3 consolidations, the 3rd preceded by a least-usage eviction
    'lt_evict': dict(H=96, W=128, nobj=2, frames=45, second=None,
                     cfg=dict(mem_every=2, max_long_term_elements=80, num_prototypes=32)),
    # a second annotated object arrives at frame 7 -> two buckets, both consolidate
    'two_buckets': dict(H=96, W=128, nobj=2, frames=30, second=(7, 5),
                        cfg=dict(mem_every=2, max_long_term_elements=300, num_prototypes=32)),
    # working memory only (BASELINE config 2 style), size needing pad on both axes
    'no_lt': dict(H=100, W=150, nobj=3, frames=12, second=None,
                  cfg=dict(enable_long_term=False, enable_long_term_count_usage=False, mem_every=3)),
    # five objects, default flags
    'five_obj': dict(H=90, W=130, nobj=5, frames=9, second=None, cfg=dict()),
}


def run_scenario(make_core: Callable[[Dict], object], sc: Dict, device: str = 'cpu',
                 on_frame: Callable = None, perturb: Callable = None) -> List[torch.Tensor]:
    """Returns the per-frame `step` outputs ([no+1,H,W] probabilities, on CPU)."""
    cfg = synth.base_config(**sc['cfg'])
    core = make_core(cfg)
    stream = synth.FrameStream(sc['H'], sc['W'], seed=1)
    mask0 = synth.box_mask(sc['H'], sc['W'], sc['nobj'])
    objs = list(range(1, sc['nobj'] + 1))
    outs = []
    for t in range(sc['frames']):
        img = stream.next()
        if perturb is not None:
            img = perturb(img)
        img = img.to(device)
        end = (t == sc['frames'] - 1)
        if t == 0:
            p = core.step(img, mask0.to(device), objs, end=end)
        elif sc['second'] is not None and t == sc['second'][0]:
            oid = sc['second'][1]
            m = torch.zeros(sc['H'], sc['W'], dtype=torch.long)
            m[sc['H'] // 2:, :sc['W'] // 4] = oid
            p = core.step(img, m.to(device), [oid], end=end)
        else:
            p = core.step(img, end=end)
        outs.append(p.detach().float().cpu())
        if on_frame is not None:
            on_frame(t, core)
    return outs, core
