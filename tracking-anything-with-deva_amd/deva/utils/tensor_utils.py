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
    if extra_h == 0 and extra_w == 0:
        return in_img, pad  # nothing to add (e.g. the input head already padded): no copy
    if in_img.is_cuda and in_img.is_contiguous() and in_img.element_size() in (1, 4, 8):
        from deva.hip import ops  # one launch instead of ATen's fill + copy (every frame of a 854-wide clip passes here)
        return ops.pad2d(in_img, pad), pad
    return F.pad(in_img, pad), pad


def unpad(img: torch.Tensor, pad: Iterable[int]) -> torch.Tensor:
    """crop what pad_divide_by added from the last two dims (2-D to 5-D inputs)"""
    if not 2 <= img.dim() <= 5:
        raise NotImplementedError
    left, right, top, bottom = pad
    h, w = img.shape[-2:]
    return img[..., top:h - bottom, left:w - right]


def network_input_size(height: int, width: int, min_side: int, antialias: bool = True) -> Tuple[int, int]:
    """output size of the reference's resize for a frame of height x width:
    antialias=True  -- torchvision `Resize(size)` of the dataset readers (video_reader.py:139-144,
                       detection_video_reader.py:63-71): the shorter side becomes exactly `min_side`, the
                       longer one int(min_side * long / short);
    antialias=False -- the demo's own rule (demo_utils.py:10-19): int(h * scale), int(w * scale) with
                       scale = min_side / min(h, w) (which can fall one short of `min_side`)."""
    if min_side <= 0:
        return height, width
    if antialias:
        short, long = (width, height) if width <= height else (height, width)
        if short == min_side:
            return height, width
        new_short, new_long = min_side, int(min_side * long / short)
        return (new_long, new_short) if width <= height else (new_short, new_long)
    scale = min_side / min(height, width)
    return int(height * scale), int(width * scale)


def frame_to_network_input(image_u8, min_side: int = -1, *, antialias: bool = True, pad_to: int = 0):
    """Device-side input head (not part of the reference's interface): a decoded
    uint8 H*W*3 frame (numpy array or tensor) -> ImageNet-normalised fp32 3*H'*W' on the HIP device with
    the shorter side resized to `min_side` (<= 0: original size; size rule: `network_input_size`), in one
    kernel.  antialias=True is the dataset readers' transform, antialias=False the demo's
    `get_input_frame_for_deva`.  Only the uint8 frame crosses PCIe.

    pad_to > 0 also fuses `pad_divide_by(image, pad_to)`: returns (padded image, (left, right, top, bottom));
    `DEVAInferenceCore.step` then finds nothing left to pad and the caller crops with `unpad(prob, pad)`."""
    from deva.hip import ops
    if not torch.is_tensor(image_u8):
        image_u8 = torch.from_numpy(image_u8)
    if not image_u8.is_cuda:
        image_u8 = image_u8.cuda()
    h, w = image_u8.shape[:2]
    oh, ow = network_input_size(h, w, min_side, antialias)
    size = None if (oh, ow) == (h, w) else (oh, ow)
    if pad_to <= 0:
        return ops.input_head(image_u8.contiguous(), size, antialias=antialias)
    extra_h, extra_w = (-oh) % pad_to, (-ow) % pad_to
    pad = (extra_w // 2, extra_w - extra_w // 2, extra_h // 2, extra_h - extra_h // 2)
    return ops.input_head(image_u8.contiguous(), size, antialias=antialias, pad=pad), pad
