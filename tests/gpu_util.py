"""helpers of the -m gpu tests: run a real HIP op and its CPU contract (tests/emu_ops.py) on the
same seeded inputs and compare."""
import os

import torch

import emu_ops
from deva.hip import ops


def net_config(**over):
    """config of the networks the -m gpu tests build: workload.synth.base_config, and with DEVA_TEST_F16_SPLIT=1 in the
    environment the SAME tests run with --f16_split (fp32-accurate convolutions on the f16 matrix pipes): every parity
    gate is then held, with unchanged bounds, under that mode (profiles/r05/tests_split/)"""
    from workload import synth
    cfg = synth.base_config()
    mode = os.environ.get('DEVA_TEST_F16_SPLIT')  # '1': value encoder + mask decoder; 'all': the key encoder too
    if mode in ('1', 'all') and not over.get('amp'):
        cfg['f16_split'] = True
        cfg['f16_split_key_encoder'] = mode == 'all'
    cfg.update(over)
    return cfg


def dev():
    assert torch.cuda.is_available(), 'the -m gpu tests need a HIP device'
    return torch.device('cuda:0')


def to_dev(x):
    if isinstance(x, torch.Tensor):
        return x.to(dev())
    if isinstance(x, ops.PackedConv):
        return ops.PackedConv(x.weight.to(dev()), None if x.bias is None else x.bias.to(dev()), x.cin, x.cout,
                              x.cout_pad, x.kh, x.kw, x.k_layout, None if x.weight_f16 is None else x.weight_f16.to(dev()),
                              None if x.weight_split is None else x.weight_split.to(dev()), x.split_scale_log2,
                              None if x.weight_wino is None else x.weight_wino.to(dev()))
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
