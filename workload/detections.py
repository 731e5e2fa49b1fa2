"""Tracker-consistent synthetic detections (BASELINE configs[2]) and the recording hooks that define a clip.

INPUT GENERATION ONLY (see workload/__init__.py).  Nothing here imports the oracle or the package: the
recorders receive the module / objects they patch from the caller (tests/, bench.py).
"""
import contextlib
from typing import Callable, Dict, List

import torch


class ConsistentDetector:
    """Synthetic detector that AGREES with the tracker (BASELINE configs[2], SURVEY.md §8d: VIPSeg-style
    clips bring ~8 segments per detection, most of them re-detections of tracked objects).

    With recipe weights the propagated masks are unrelated to any fixed box, so a detection drawn at a
    fixed place never reaches IoU 0.5 with a tracked object and every segment would spawn a new object.
    This generator is therefore a pure function of the tracker's own forward hard mask (object ids):
    * up to `segments - new_per_frame` tracked objects flagged "stable" ((id // 10) % 4 == 1, table order)
      are re-detected as their forward region minus every 8th 8-row band (IoU 0.875 with the forward
      mask -> matched and merged, segment_merging.py:28-86);
    * the remaining segments are new boxes (ids 10, 20, 30, ... in order of creation, alternating thing /
      stuff / untyped) -> new objects, a new memory bucket per detection (kv_memory_store.py:66-89);
    * every other tracked object is never re-detected -> its poke count grows and it is purged after
      `max_missed_detection_count` misses (object_manager.py:89-110, memory_manager.purge_except).
    The test / bench harness feeds `forward` from a hook around the merge step of the run that defines
    the clip, records the detections and replays them through the public `incorporate_detection`."""

    def __init__(self, height: int, width: int, segments: int = 8, new_per_frame: int = 2):
        self.h, self.w, self.segments, self.new_per_frame = height, width, segments, new_per_frame
        self.counter = 0

    @staticmethod
    def is_stable(obj_id: int) -> bool:
        return (obj_id // 10) % 4 == 1

    def __call__(self, forward: torch.Tensor, live: List[Dict], t: int):
        """forward: H*W long, tracker's forward mask in OBJECT ids (0 = background; all zeros on the first
        frame); live: [{id, isthing}] tracked objects in table order -> (H*W long detection, segments_info)"""
        h, w = self.h, self.w
        assert tuple(forward.shape) == (h, w)
        det = torch.zeros(h, w, dtype=torch.long)
        matched = []
        rows = torch.arange(h).view(-1, 1)
        band = ((rows + 3 * t) // 8) % 8 == 0
        for rec in live:
            if len(matched) >= self.segments - self.new_per_frame:
                break
            if not self.is_stable(rec['id']):
                continue
            region = (forward == rec['id'])
            if int(region.sum()) < (h * w) // 2000:
                continue
            matched.append((rec, region & ~band))
        info = []
        bh, bw = h // 6, w // 8
        for _ in range(self.segments - len(matched)):
            c = self.counter
            self.counter += 1
            y0, x0 = (c * 104729 + 17) % (h - bh), (c * 7919 + 5) % (w - bw)
            new_id = 10 * (c + 1)
            det[y0:y0 + bh, x0:x0 + bw] = new_id
            isthing = (True, False, None)[c % 3]
            info.append(dict(id=new_id, category_id=None if isthing is None else c % 5 + 1, isthing=isthing))
        for rec, region in matched:  # re-detections on top: their IoU with the forward mask stays 0.875
            det[region] = 100000 + rec['id']
            info.append(dict(id=100000 + rec['id'], category_id=None if rec['isthing'] is None else 9,
                             isthing=rec['isthing']))
        return det, info


@contextlib.contextmanager
def record_on_package(core, detector: ConsistentDetector, make_info: Callable, recorded: Dict[int, tuple],
                      frame_of: Callable[[], int]):
    """While active, `core.incorporate_detection` ignores the detection it is handed and merges
    `detector(forward mask of THIS run)` instead: a hook around `match_and_merge`, the only consumer of the
    detection (inference_core.py:172-177), of whichever `deva` tree is importable (the package; the
    reference in tests/golden/make_golden.py).  The generated (mask, segments_info) pairs are stored in
    `recorded[frame]` for replay through the public interface.  core: the DEVAInferenceCore being driven
    (its `.pad` is the current frame's padding)."""
    import deva.inference.inference_core as ic
    import deva.inference.segment_merging as sm
    from deva.utils.tensor_utils import pad_divide_by, unpad
    real = sm.match_and_merge

    def generating(our_mask, new_mask, object_manager, new_segments_info, **kw):
        lut = torch.zeros(len(object_manager.tmp_id_to_obj) + 1, dtype=torch.long)
        live = []
        for tmp, obj in object_manager.tmp_id_to_obj.items():
            lut[tmp] = obj.id
            live.append(dict(id=obj.id, isthing=obj.isthing))
        forward = lut[unpad(our_mask, core.pad).long().cpu()]
        t = frame_of()
        det, info = detector(forward, live, t)
        recorded[t] = (det, info)
        return real(our_mask, pad_divide_by(det.to(our_mask.device), 16)[0], object_manager,
                    [make_info(**i) for i in info], **kw)

    bound_at_import = getattr(ic, 'match_and_merge', None) is real  # the reference binds the name at import
    sm.match_and_merge = generating
    if bound_at_import:
        ic.match_and_merge = generating
    try:
        yield recorded
    finally:
        sm.match_and_merge = real
        if bound_at_import:
            ic.match_and_merge = real


@contextlib.contextmanager
def record_on_oracle(oracle_module, detector: ConsistentDetector, recorded: Dict[int, tuple],
                     frame_of: Callable[[], int], pad_of: Callable[[], tuple]):
    """The same hook around `merge_detection` of the CPU oracle (tests only: the caller passes the oracle
    module in).  pad_of() -> the pad tuple of the current frame (oracle.pad_to_multiple)."""
    O = oracle_module
    real = O.merge_detection

    def generating(forward, detected, table, segments, history, **kw):
        lut = torch.tensor([0] + [rec['id'] for rec in table], dtype=torch.long)
        live = [dict(id=rec['id'], isthing=rec['isthing']) for rec in table]
        t = frame_of()
        det, info = detector(lut[O.unpad(forward, pad_of()).long()], live, t)
        recorded[t] = (det, info)
        return real(forward, O.pad_to_multiple(det)[0], table, info, history, **kw)

    O.merge_detection = generating
    try:
        yield recorded
    finally:
        O.merge_detection = real
