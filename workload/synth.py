"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

INPUT GENERATION ONLY (see workload/__init__.py).  Everything is generated on the CPU with
explicitly seeded `torch.Generator`s so the same tensors can be rebuilt on any box.
Recipes follow SURVEY.md §8d ("Synthetic inputs").
"""
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


def base_config(**overrides) -> Dict:
    """Hot-path keys and defaults of deva/inference/eval_args.py:17-56 (+ the per-video flag of
    evaluation/eval_vos.py:127-130)."""
    cfg = dict(pix_feat_dim=512, key_dim=64, value_dim=512, enable_long_term=True,
               enable_long_term_count_usage=True, max_mid_term_frames=10, min_mid_term_frames=5,
               max_long_term_elements=10000, num_prototypes=128, top_k=30, mem_every=5,
               chunk_size=-1, size=480, amp=False)
    cfg.update(overrides)
    return cfg


class FrameStream:
    """Temporally coherent smooth-noise frames: img_t = 0.9 img_{t-1} + 0.1 fresh, already
    'ImageNet-normalised' (zero-mean unit-ish variance)."""

    def __init__(self, height: int, width: int, seed: int = 1):
        self.h, self.w = height, width
        self.g = torch.Generator(device='cpu').manual_seed(seed)
        self.img = self._fresh()

    def _fresh(self) -> torch.Tensor:
        lo = torch.randn(1, 3, max(self.h // 8, 1), max(self.w // 8, 1), generator=self.g)
        lo = F.interpolate(lo, size=(self.h, self.w), mode='bilinear', align_corners=False)[0]
        return lo + 0.3 * torch.randn(3, self.h, self.w, generator=self.g)

    def next(self) -> torch.Tensor:
        out = self.img
        self.img = 0.9 * self.img + 0.1 * self._fresh()
        return out


def box_mask(height: int, width: int, num_objects: int, first_id: int = 1) -> torch.Tensor:
    """Index mask with `num_objects` overlapping-free staggered rectangles, ids first_id.."""
    m = torch.zeros(height, width, dtype=torch.long)
    for o in range(num_objects):
        y0, x0 = (o * height) // (num_objects + 1), (o * width) // (num_objects + 1)
        m[y0:y0 + height // 3, x0:x0 + width // 3] = first_id + o
    return m


def affinity_inputs(n: int, hw: int, ck: int = 64, seed: int = 0, key_scale: float = 1.0):
    """Kernel-level inputs: mk~N(0,1)*scale, ms~U(1,2), qk~N(0,1)*scale, qe~U(0,1)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    mk = torch.randn(ck, n, generator=g) * key_scale
    ms = torch.rand(1, n, generator=g) + 1.0
    qk = torch.randn(ck, hw, generator=g) * key_scale
    qe = torch.rand(ck, hw, generator=g)
    return mk, ms, qk, qe


def value_inputs(num_objects: int, cv: int, n: int, seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device='cpu').manual_seed(seed + 7919)
    return torch.randn(num_objects, cv, n, generator=g)


def stage_inputs(height: int, width: int, num_objects: int, seed: int = 3):
    """Teacher-forcing inputs for encode_mask / segment at a given (padded) frame size."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    h, w = height // 16, width // 16
    masks = torch.rand(1, num_objects, height, width, generator=g)
    masks = masks / masks.sum(1, keepdim=True).clamp(min=1.0) * 0.9
    sensory = torch.randn(1, num_objects, 512, h, w, generator=g) * 0.5
    readout = torch.randn(1, num_objects, 512, h, w, generator=g) * 0.5
    return masks, sensory, readout


def prefill_bank(n: int, objects: List[int], seed: int = 1, ck: int = 64, cv: int = 512):
    """long-term bank contents mk~N(0,1) [ck,n], ms~U(1,2) [1,n],
    values~N(0,1) {obj: [cv,n]}, to be appended through the store's own `add(key, values, shrinkage,
    None, <bucket 0>)` after the first frame has created bucket 0."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    key = torch.randn(ck, n, generator=g)
    shr = torch.rand(1, n, generator=g) + 1.0
    values = {o: torch.randn(cv, n, generator=g) for o in objects}
    return key, shr, values


def detection_frame(height: int, width: int, t: int, segments: int = 1):
    """Synthetic precomputed detection of frame t (BASELINE configs[2], eval_with_detections style):
    an index mask with `segments` boxes that drift 2 px per frame, ids 10, 20, ..., and their
    segments_info dicts (alternating thing / stuff, category = id // 10)."""
    m = torch.zeros(height, width, dtype=torch.long)
    info = []
    bh, bw = height // 4, width // (2 * segments + 2)
    for s in range(segments):
        y0 = (height // 8) + (s % 2) * (height // 2)
        x0 = (2 * s + 1) * bw // 1 + 2 * t
        m[y0:y0 + bh, x0:x0 + bw] = 10 * (s + 1)
        info.append(dict(id=10 * (s + 1), category_id=s + 1, isthing=(s % 2 == 0)))
    return m, info

