"""The frame loops of the reference's two evaluation drivers, restated against the PUBLIC interface only
(`get_model_and_config`-style network, `DEVAInferenceCore`, `ObjectInfo`), so that the same code drives

* the reference on the CPU (tests/golden/make_golden.py -> tests/golden/driver_semionline.npz), and
* this package on libdeva_hip.so (tests/test_gpu_i_drivers.py, `-m gpu`; the GPU boxes carry no reference
  checkout, so the UNCHANGED scripts cannot run there -- tests/test_gpu_h_reference_drivers.py keeps that leg for
  boxes that have one).

`eval_vos_loop`   evaluation/eval_vos.py:133-184: DataLoader-shaped CPU batches -> `.cuda()` -> `processor.step` ->
                  (F.interpolate) -> argmax -> `tmp_to_obj_cls` -> per-frame event pair + synchronize -> index mask to
                  the saver (a thread, like ResultSaver).
`semionline_loop` evaluation/eval_with_detections.py:150-297, temporal_setting == 'semionline': frames are buffered
                  until `num_voting_frames` detections are there, `vote_in_temporary_buffer(keyframe_selection='first')`
                  -> `incorporate_detection` on the first buffered frame, plain `step` on the rest of the buffer,
                  `clear_buffer`, propagation until the next voting frame.
Nothing here imports the oracle or the reference; `device` = 'cpu' (reference / emulated ops) or 'cuda'."""
import queue
import threading
from typing import Callable, Dict, List

import numpy as np
import torch
import torch.nn.functional as F

IM_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)  # deva/dataset/utils.py:8 (im_normalization)
IM_STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


def to_network_input(frame_u8_hwc: torch.Tensor) -> torch.Tensor:
    """ToTensor + im_normalization of the readers (video_reader.py:133-150), on the CPU like a DataLoader worker"""
    x = frame_u8_hwc.permute(2, 0, 1).float().div(255)
    return (x - IM_MEAN) / IM_STD


class FrameInfo:
    """the fields of deva/inference/frame_utils.py:7-30"""

    def __init__(self, image, mask, segments_info, ti, info):
        self.image, self.mask, self.segments_info, self.ti, self.info = image, mask, segments_info, ti, info

    name = property(lambda self: self.info['frame'][0])
    shape = property(lambda self: self.info['shape'])
    save_needed = property(lambda self: self.info['save'][0])
    path_to_image = property(lambda self: self.info['path_to_image'][0])


class MaskSaver:
    """stands in for ResultSaver (result_utils.py:104-163): a daemon thread that receives `.cpu()` copies"""

    def __init__(self):
        self.masks: Dict[str, torch.Tensor] = {}
        self._q: 'queue.Queue' = queue.Queue()
        self._t = threading.Thread(target=self._work, daemon=True)
        self._t.start()

    def _work(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            name, mask = item
            self.masks[name] = mask.to(torch.int32)

    def save(self, name: str, index_mask: torch.Tensor):
        self._q.put((name, index_mask.detach().cpu()))

    def end(self) -> Dict[str, torch.Tensor]:
        self._q.put(None)
        self._t.join()
        return self.masks


def _sync(device):
    if torch.device(device).type == 'cuda':
        torch.cuda.synchronize()


def _events(device):
    if torch.device(device).type != 'cuda':
        return None, None
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def _count_usage(config: Dict, vid_length: int) -> Dict:
    """eval_vos.py:126-131 / eval_with_detections.py:138-142: no need to count usage for LT if the video is not that long"""
    config = dict(config)
    config['enable_long_term_count_usage'] = bool(
        config['enable_long_term'] and
        (vid_length / (config['max_mid_term_frames'] - config['min_mid_term_frames']) * config['num_prototypes'])
        >= config['max_long_term_elements'])
    return config


def eval_vos_loop(make_processor: Callable[[Dict], object], config: Dict, frames_u8: torch.Tensor, annotation: torch.Tensor,
                  device='cuda', out_size=None):
    """frames_u8 [T,H,W,3] uint8, annotation [H,W] uint8 palette indices of frame 0 -> ({frame name: index mask},
    seconds of the per-frame timed regions).  out_size != None exercises the need_resize branch."""
    vid_length = frames_u8.shape[0]
    processor = make_processor(_count_usage(config, vid_length))
    saver = MaskSaver()
    total_time, first_mask_loaded = 0.0, False
    labels = torch.unique(annotation)
    labels = labels[labels != 0]
    for ti in range(vid_length):
        # what DataLoader(vid_reader, batch_size=1) hands over: CPU tensors with a leading batch dimension
        data = {'rgb': to_network_input(frames_u8[ti]).unsqueeze(0),
                'info': {'frame': [f'{ti:05d}.jpg'], 'shape': [torch.tensor([frames_u8.shape[1]]), torch.tensor([frames_u8.shape[2]])],
                         'need_resize': torch.tensor([out_size is not None])}}
        if ti == 0:
            data['mask'] = annotation.unsqueeze(0)
            data['valid_labels'] = labels.unsqueeze(0)
        image = data['rgb'].to(device)[0]
        mask = data.get('mask')
        if mask is not None:
            mask = mask.to(device)[0]
        valid_labels = data.get('valid_labels')
        if valid_labels is not None:
            valid_labels = valid_labels.tolist()[0]
        info = data['info']
        need_resize = bool(info['need_resize'][0])
        start, end = _events(device)
        if start is not None:
            start.record()
        if not first_mask_loaded:
            if mask is None:
                continue
            first_mask_loaded = True
        prob = processor.step(image, mask, valid_labels, end=(ti == vid_length - 1))
        if need_resize:
            prob = F.interpolate(prob.unsqueeze(1), out_size, mode='bilinear', align_corners=False)[:, 0]
        out_mask = torch.argmax(prob, dim=0)
        out_mask = processor.object_manager.tmp_to_obj_cls(out_mask)
        if end is not None:
            end.record()
        _sync(device)
        if start is not None:
            total_time += start.elapsed_time(end) / 1000
        saver.save(info['frame'][0][:-4], out_mask)
    return saver.end(), total_time


def semionline_loop(make_processor: Callable[[Dict], object], config: Dict, frames_u8: torch.Tensor, detections: List, make_info: Callable,
                    num_voting_frames=3, detection_every=5, device='cuda', seed=0):
    """detections[t] = (index mask [H,W] long, [dict(id, category_id, isthing, score)]) for EVERY frame (the driver's
    reader delivers a detection with every frame; only the voting windows use them).
    -> ({frame name: index mask in object ids}, [object ids alive at the end])
    seed: in long-id mode the object manager re-draws ids below 256 from np.random (object_manager.py:40-50); the
    drivers do not seed it, a comparison against stored masks has to."""
    np.random.seed(seed)
    vid_length = frames_u8.shape[0]
    processor = make_processor(_count_usage(config, vid_length))
    saver = MaskSaver()
    next_voting_frame = num_voting_frames - 1
    processor.enabled_long_id()

    def save(prob, name):
        out_mask = torch.argmax(prob, dim=0)
        saver.save(name[:-4], processor.object_manager.tmp_to_obj_cls(out_mask))

    for ti in range(vid_length):
        det_mask, det_info = detections[ti]
        image = to_network_input(frames_u8[ti]).unsqueeze(0).to(device)[0]
        mask = det_mask.unsqueeze(0).to(device)[0]
        info = {'frame': [f'{ti:05d}.jpg'], 'shape': None, 'need_resize': [False], 'save': [True], 'path_to_image': [None]}
        segments_info = [make_info(**i) for i in det_info]  # convert_json_dict_to_objects_info (result_utils.py)
        frame_info = FrameInfo(image, mask, segments_info, ti, info)
        if ti + num_voting_frames > next_voting_frame:
            processor.add_to_temporary_buffer(frame_info)
            if ti == next_voting_frame:
                first = processor.frame_buffer[0]
                _, vmask, new_segments_info = processor.vote_in_temporary_buffer(keyframe_selection='first')
                prob = processor.incorporate_detection(first.image, vmask, new_segments_info)
                next_voting_frame += detection_every
                if next_voting_frame >= vid_length:
                    next_voting_frame = vid_length + num_voting_frames
                _sync(device)
                save(prob, first.name)
                for fi in processor.frame_buffer[1:]:
                    prob = processor.step(fi.image, None, None, end=(fi.ti == vid_length - 1))
                    _sync(device)
                    save(prob, fi.name)
                processor.clear_buffer()
        else:
            prob = processor.step(image, None, None, end=(ti == vid_length - 1))
            _sync(device)
            save(prob, info['frame'][0])
    alive = [int(o.id) for o in processor.object_manager.obj_to_tmp_id]
    return saver.end(), alive


def semionline_clip(H=96, W=128, frames=13, seed=21):
    """synthetic clip for the semi-online loop: temporally coherent uint8 frames and, for every frame, a detection with
    a thing box drifting right, a stuff box that is missed on some frames, and a spurious box on frame 0"""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(H, W, 3, generator=g)
    imgs, dets = [], []
    for t in range(frames):
        base = (0.9 * base + 0.1 * torch.rand(H, W, 3, generator=g)).clamp(0, 1)
        imgs.append((base * 255).to(torch.uint8))
        m = torch.zeros(H, W, dtype=torch.long)
        info = []
        m[12:52, 10 + 2 * t:58 + 2 * t] = 7 + 10 * t
        info.append(dict(id=7 + 10 * t, category_id=2, isthing=True, score=0.9))
        if t % 4 != 1:
            m[56:92, 66:122] = 3 + 10 * t
            info.append(dict(id=3 + 10 * t, category_id=5, isthing=False, score=0.7))
        if t == 0:
            m[60:90, 4:30] = 5
            info.append(dict(id=5, category_id=2, isthing=True, score=0.4))
        dets.append((m, info))
    return torch.stack(imgs), dets
