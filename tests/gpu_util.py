"""helpers of the -m gpu tests: run a real HIP op and its CPU contract (tests/emu_ops.py) on the
same seeded inputs and compare."""
import torch

import emu_ops
from deva.hip import ops


def dev():
    assert torch.cuda.is_available(), 'the -m gpu tests need a HIP device'
    return torch.device('cuda:0')


def to_dev(x):
    if isinstance(x, torch.Tensor):
        return x.to(dev())
    if isinstance(x, ops.PackedConv):
        return ops.PackedConv(x.weight.to(dev()), None if x.bias is None else x.bias.to(dev()), x.cin, x.cout,
                              x.cout_pad, x.kh, x.kw, x.k_layout, None if x.weight_f16 is None else x.weight_f16.to(dev()))
    if isinstance(x, dict):
        return {k: to_dev(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(to_dev(v) for v in x)
    return x


def max_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    b = b.detach().float().cpu()
    return max_err(a, b) / max(1e-12, b.abs().max().item())


def rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale
