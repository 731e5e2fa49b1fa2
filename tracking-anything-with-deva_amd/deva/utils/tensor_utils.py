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


def frame_to_network_input(image_u8, min_side: int = -1, *, antialias: bool = True) -> torch.Tensor:
    """Device-side input head (not part of the reference's interface): a decoded
    uint8 H*W*3 frame (numpy array or tensor) -> ImageNet-normalised fp32 3*H'*W' on the HIP device with
    the shorter side resized to `min_side` (<= 0: original size), in one kernel.  antialias=True is the
    dataset readers' transform (video_reader.py:139-144), antialias=False the demo's
    `get_input_frame_for_deva` (demo_utils.py:10-19).  Only the uint8 frame crosses PCIe."""
    from deva.hip import ops
    if not torch.is_tensor(image_u8):
        image_u8 = torch.from_numpy(image_u8)
    if not image_u8.is_cuda:
        image_u8 = image_u8.cuda()
    h, w = image_u8.shape[:2]
    size = None
    if min_side > 0:
        scale = min_side / min(h, w)
        size = (int(h * scale), int(w * scale))
    return ops.input_head(image_u8.contiguous(), size, antialias=antialias)
