"""Command-line surface shared by the evaluation drivers and construction of the HIP network
(interface of deva/inference/eval_args.py:7-71: `add_common_eval_args`, `get_model_and_config`).

The flags are data here: one table row per flag (name, type or None for a switch, default, help),
so that the surface the drivers rely on can be read -- and checked against the reference -- at a
glance."""
from argparse import ArgumentParser
from typing import Dict, Tuple

import torch

# (flag, type | None = store_true switch, default, help)
_FLAGS = (
    ('model', str, './saves/DEVA-propagation.pth', 'checkpoint with the 420 propagation tensors'),
    ('output', str, None, 'where the driver writes its results'),
    ('save_all', None, False, 'Save all frames'),
    ('amp', None, False, 'fp16 operands / fp32 accumulation in the value encoder and the mask decoder (key encoder and '
                         'memory read stay fp32); off: everything fp32, the parity target'),
    ('f16_split', None, False, 'extension: fp32-accurate convolutions on the f16 matrix pipes (hi/lo fp16 split of both '
                               'operands, fp32 accumulation) in the value encoder and the mask decoder; held to the fp32 '
                               'parity bounds, 2.5-3x the fp32 rate of those layers.  Error per product: 2^-21 |x w| + 2^-25 |w| '
                               '(absolute floor: activations are not pre-scaled; layers whose inputs are all << 0.1 lose bits)'),
    ('no_winograd', None, False, 'extension: keep the big 3x3 layers (value encoder / mask decoder; key encoder from 1080p up) on the '
                                 'direct fp32 kernels (default: Winograd F(2x2, 3x3) on the fp32 matrix pipes, 2.25x fewer '
                                 'multiply-adds, same parity gates)'),
    ('f16_split_key_encoder', None, False, 'extension, with --f16_split: the key encoder on the split kernels too (the memory '
                                           "read's inputs then move by fp32 round-off, as under any change of accumulation order)"),
    # network widths (C^k, C^v, pixel feature)
    ('key_dim', int, 64, None),
    ('value_dim', int, 512, None),
    ('pix_feat_dim', int, 512, None),
    # memory schedule (XMem notation)
    ('disable_long_term', None, False, 'working memory only'),
    ('max_mid_term_frames', int, 10, 'T_max in XMem, decrease to save memory'),
    ('min_mid_term_frames', int, 5, 'T_min in XMem, decrease to save memory'),
    ('max_long_term_elements', int, 10000, 'LT_max in XMem, increase if objects disappear for a long time'),
    ('num_prototypes', int, 128, 'P in XMem'),
    ('top_k', int, 30, 'memory tokens in the softmax support of a query'),
    ('mem_every', int, 5, 'r in XMem. Increase to improve running speed.'),
    ('chunk_size', int, -1, 'Number of objects to process in parallel as a batch; -1 for unlimited. '
                            'Set to a small number to save memory.'),
    ('size', int, 480, 'Resize the shorter side to this size. -1 to use original resolution. '),
)


def add_common_eval_args(parser: ArgumentParser) -> None:
    for flag, kind, default, doc in _FLAGS:
        if kind is None:
            parser.add_argument(f'--{flag}', action='store_true', help=doc)
        elif kind is str:
            parser.add_argument(f'--{flag}', default=default, help=doc)
        else:
            parser.add_argument(f'--{flag}', type=kind, default=default, help=doc)


def get_model_and_config(parser: ArgumentParser) -> Tuple['torch.nn.Module', Dict, object]:
    """-> (network on the current HIP device in eval mode, config dict, parsed args)"""
    from deva.model.network import DEVA
    args = parser.parse_args()
    config = dict(vars(args), enable_long_term=not args.disable_long_term)
    network = DEVA(config).cuda().eval()
    if args.model is None:
        print('No model loaded.')
    else:
        network.load_weights(torch.load(args.model))
    return network, config, args
