"""Common evaluation flags and model construction (interface of deva/inference/eval_args.py:7-71)."""
from argparse import ArgumentParser

import torch

from deva.model.network import DEVA


def add_common_eval_args(parser: ArgumentParser):
    parser.add_argument('--model', default='./saves/DEVA-propagation.pth')
    parser.add_argument('--output', default=None)
    parser.add_argument('--save_all', action='store_true', help='Save all frames')
    parser.add_argument('--amp', action='store_true')

    # model dimensions
    parser.add_argument('--key_dim', type=int, default=64)
    parser.add_argument('--value_dim', type=int, default=512)
    parser.add_argument('--pix_feat_dim', type=int, default=512)

    # long-term memory
    parser.add_argument('--disable_long_term', action='store_true')
    parser.add_argument('--max_mid_term_frames', type=int, default=10,
                        help='T_max in XMem, decrease to save memory')
    parser.add_argument('--min_mid_term_frames', type=int, default=5,
                        help='T_min in XMem, decrease to save memory')
    parser.add_argument('--max_long_term_elements', type=int, default=10000,
                        help='LT_max in XMem, increase if objects disappear for a long time')
    parser.add_argument('--num_prototypes', type=int, default=128, help='P in XMem')

    parser.add_argument('--top_k', type=int, default=30)
    parser.add_argument('--mem_every', type=int, default=5,
                        help='r in XMem. Increase to improve running speed.')
    parser.add_argument('--chunk_size', type=int, default=-1,
                        help='Number of objects to process in parallel as a batch; -1 for unlimited. '
                        'Set to a small number to save memory.')
    parser.add_argument('--size', type=int, default=480,
                        help='Resize the shorter side to this size. -1 to use original resolution. ')


def get_model_and_config(parser: ArgumentParser):
    args = parser.parse_args()
    config = vars(args)
    config['enable_long_term'] = not config['disable_long_term']

    network = DEVA(config).cuda().eval()
    if args.model is not None:
        network.load_weights(torch.load(args.model))
    else:
        print('No model loaded.')
    return network, config, args
