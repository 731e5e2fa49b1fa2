"""Frame padding helpers with the reference's interface (deva/utils/tensor_utils.py:7-48)."""
from typing import Iterable, Tuple

import torch
import torch.nn.functional as F


def pad_divide_by(in_img: torch.Tensor, d: int) -> Tuple[torch.Tensor, Tuple[int, int, int, int]]:
    """zero-pad the last two dims up to multiples of d; the odd pixel goes to the bottom/right.
    Returns the padded tensor and (left, right, top, bottom)."""
    h, w = in_img.shape[-2:]
    extra_h, extra_w = (-h) % d, (-w) % d
    pad = (extra_w // 2, extra_w - extra_w // 2, extra_h // 2, extra_h - extra_h // 2)
    return F.pad(in_img, pad), pad


def unpad(img: torch.Tensor, pad: Iterable[int]) -> torch.Tensor:
    """crop what pad_divide_by added from the last two dims (2-D to 5-D inputs)"""
    if not 2 <= img.dim() <= 5:
        raise NotImplementedError
    left, right, top, bottom = pad
    h, w = img.shape[-2:]
    return img[..., top:h - bottom, left:w - right]
