"""Deterministic synthetic weights for the DEVA propagation network.

INPUT GENERATION ONLY (see workload/__init__.py): only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this package.

The real checkpoint (`DEVA-propagation.pth`, 420 tensors / 277 MB fp32,
reference `scripts/download_models.sh:1`) is a network download and cannot be
committed.  Instead every tensor of the state_dict is filled from a recipe keyed
on (seed, tensor name), so that the reference (in the fixture generator), the
oracle restatement and the HIP runtime all load bit-identical weights without
shipping them.  BatchNorm statistics are deliberately non-trivial so BN folding
is exercised.
"""
import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


# Per-tensor gains applied on top of the He-style fill.  With them the synthetic network has
# peaky (trained-like) memory affinities (top-1 weight ~0.07, 30th ~0.03) and mask logits of
# moderate magnitude instead of saturated ones; probed against the reference in the container.
GAINS = {
    'key_proj.key_proj.weight': 6.0,
    'mask_decoder.pred.weight': 0.4,
}

# Second recipe, "peaky" (VERDICT r2 item 1): same fill, sharper memory affinities and mask logits.
# A trained XMem/DEVA read is dominated by a few tokens (the 30th of the top-30 weights is negligible) and
# its probabilities are far from flat.  With the default gains the 30th weight is still ~0.4x the first, so a
# near-tie at the k-th/(k+1)-th boundary swaps a token that carries 3 % of the read-out, and 60 % of the
# pixels have a top-1/top-2 probability margin below 1e-2: "argmax-identical" is then untestable.  Key gain
# 15 makes the boundary tokens weightless (the reference's own drift under a 1e-6 input perturbation drops
# from 4e-3 to 4e-4 at 240x432; the scores grow with the gain squared: at gain 30 the best score of some
# 480p queries falls below -103 and the reference's exp() without max subtraction returns 0/0 = NaN from
# frame 4 on, gain 60 does so everywhere), prediction gain 1.0 puts 80 % of the pixels above a 1e-2 margin
# (gain 2.0 turns the random recurrent network chaotic: its self-drift grows to 4e-2) -- probed with the
# oracle, 240x432 / 3 objects / 12 frames and 480x854 / 5 objects / 7 frames.
RECIPES = {
    'default': GAINS,
    'peaky': {'key_proj.key_proj.weight': 15.0, 'mask_decoder.pred.weight': 1.0},
}


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device='cpu')
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))
    return g


def fill_tensor(name: str, shape: Tuple[int, ...], dtype: torch.dtype, seed: int,
                is_bn: bool, gains: Dict[str, float] = None) -> torch.Tensor:
    gains = GAINS if gains is None else gains
    g = _gen(seed, name)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=dtype)
    if is_bn:
        if leaf == 'weight':
            return torch.empty(shape).uniform_(0.35, 0.75, generator=g)
        if leaf == 'bias':
            return torch.empty(shape).normal_(0.0, 0.1, generator=g)
        if leaf == 'running_mean':
            return torch.empty(shape).normal_(0.0, 0.1, generator=g)
        if leaf == 'running_var':
            return torch.empty(shape).uniform_(0.5, 1.5, generator=g)
        raise KeyError(name)
    if leaf == 'weight':
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        std = math.sqrt(2.0 / fan_in)
        return torch.empty(shape).normal_(0.0, std, generator=g) * gains.get(name, 1.0)
    if leaf == 'bias':
        return torch.empty(shape).normal_(0.0, 0.05, generator=g)
    raise KeyError(name)


def make_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...], torch.dtype]],
                    seed: int = 0, recipe: str = 'default') -> Dict[str, torch.Tensor]:
    """spec: iterable of (name, shape, dtype) in state_dict order; recipe: key of RECIPES."""
    gains = RECIPES[recipe]
    spec = list(spec)
    names = {n for n, _, _ in spec}
    out = {}
    for name, shape, dtype in spec:
        prefix = name.rsplit('.', 1)[0]
        is_bn = (prefix + '.running_mean') in names
        out[name] = fill_tensor(name, tuple(shape), dtype, seed, is_bn, gains)
    return out
