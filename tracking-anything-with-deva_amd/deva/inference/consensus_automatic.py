"""`find_consensus_auto_association` -- semi-online consensus over a window of per-frame detections when
the association between the frames' segments has to be inferred (reference:
deva/inference/consensus_automatic.py:82-290; caller `DEVAInferenceCore.vote_in_temporary_buffer`,
SURVEY.md 8f #2).

What runs on the MI355X: every frame of the window is projected onto the keyframe with the fused
one-frame memory read (`spatial_alignment`), turned into a label map by the index-mask kernel, and the
pairwise-IoU table of ALL segment pairs of two frames comes from ONE joint label histogram per frame
pair (`deva_label_histogram`) followed by a single device-to-host copy of all tables -- the reference
builds `combined == label` masks and calls `.sum().item()` once per candidate pair (:206-209, one
host sync each).  The decisions (greedy IoU > 0.5 matching per thing / stuff / untyped class, :196-223),
the 0/1 programme that selects the supported, mutually non-overlapping segments (:55-79) and the
merging of their meta data (:258-272) are host logic and follow the reference step for step.

The 0/1 programme is a maximum-weight independent set on the (tiny) graph of matched pairs.  PuLP/CBC is
used when it is installed -- it is what the reference falls back to without Gurobi; otherwise the
programme is solved exactly by enumeration over its connected components (`solve_exact`).
"""
from collections import defaultdict
from typing import Dict, List, Literal, Tuple

import warnings

import numpy as np
import torch

from deva.hip import ops
from deva.inference.consensus_associated import spatial_alignment
from deva.inference.object_info import ObjectInfo
from deva.utils.tensor_utils import pad_divide_by, unpad


def solve_exact(pairwise_iou: np.ndarray, pairwise_iou_indicator: np.ndarray, total_segments: int) -> List[bool]:
    """maximise sum_j x_j * (2 * sum_i iou[i, j] - 1)  subject to  x_i + x_j <= 1 wherever
    indicator[i, j]  (consensus_automatic.py:55-79), exactly: the conflict graph only links segments of
    different frames that match each other, so its connected components have a handful of nodes and are
    enumerated."""
    weight = pairwise_iou.sum(axis=0) * 2.0 - 1.0
    conflict = [set(np.nonzero(pairwise_iou_indicator[i])[0].tolist()) - {i} for i in range(total_segments)]
    chosen = [False] * total_segments
    seen = set()
    for root in range(total_segments):
        if root in seen:
            continue
        comp, stack = [], [root]
        seen.add(root)
        while stack:
            u = stack.pop()
            comp.append(u)
            for v in conflict[u]:
                if v not in seen:
                    seen.add(v)
                    stack.append(v)
        comp.sort()
        best_val, best_set = 0.0, ()
        if len(comp) > 22:  # 4 M subsets at most
            # matches are not transitive across frame pairs: a long voting window can chain many segments into one
            # conflict component, and 2^|component| subsets would stall the host.  Greedy by weight instead (the
            # default warning filter reports this line once).  (Self-contained on purpose: the test harness executes
            # this function's source inside the reference for its reference-only leg.)
            import warnings as _warnings
            _warnings.warn(f'consensus: a conflict component of {len(comp)} segments exceeds the exact solver\'s limit '
                           '(22); falling back to a greedy selection for it', RuntimeWarning)
            for u in sorted(comp, key=lambda v: (-float(weight[v]), v)):
                if weight[u] > 0 and not any(chosen[v] for v in conflict[u]):
                    chosen[u] = True
            continue

        def search(pos: int, picked: Tuple[int, ...], value: float, best):
            if pos == len(comp):
                return (value, picked) if value > best[0] + 1e-12 else best
            u = comp[pos]
            best = search(pos + 1, picked, value, best)  # leave u out
            if weight[u] > 0 and not any(v in conflict[u] for v in picked):
                best = search(pos + 1, picked + (u,), value + float(weight[u]), best)
            return best

        best_val, best_set = search(0, (), 0.0, (best_val, best_set))
        for u in best_set:
            chosen[u] = True
    return chosen


def solve(pairwise_iou: np.ndarray, pairwise_iou_indicator: np.ndarray, total_segments: int) -> List[bool]:
    try:
        import pulp
        if not hasattr(pulp, 'LpProblem'):
            raise ImportError
    except ImportError:
        return solve_exact(pairwise_iou, pairwise_iou_indicator, total_segments)
    m = pulp.LpProblem('prob', pulp.LpMaximize)
    x = pulp.LpVariable.dicts('x', range(total_segments), cat=pulp.LpBinary)
    m += pulp.LpAffineExpression([(x[i], float(pairwise_iou[:, i].sum()) * 2 - 1) for i in range(total_segments)])
    for i in range(total_segments):
        for j in range(i + 1, total_segments):
            if pairwise_iou_indicator[i, j]:
                m += pulp.LpConstraint(pulp.LpAffineExpression([(x[i], 1), (x[j], 1)]), pulp.LpConstraintLE,
                                       f'{i}-{j}', 1)
    m.solve(pulp.PULP_CBC_CMD(msg=0))
    out = [False] * total_segments
    for v in m.variables():
        out[int(v.name[2:])] = bool(v.varValue and v.varValue > 0.5)
    return out


def pairwise_tables(label_maps: List[torch.Tensor], counts: List[int]) -> Dict[Tuple[int, int], np.ndarray]:
    """joint label histograms of every frame pair (i < j) of the window: tables[(i, j)][a, b] = number of
    pixels with local label a (0 = background) in frame i's projected map and label b in frame j's (last
    column: b = 0).  One kernel launch per pair, ONE device-to-host copy for all of them."""
    pairs, flat = [], []
    for i in range(len(label_maps)):
        for j in range(i + 1, len(label_maps)):
            if label_maps[i] is None or label_maps[j] is None:
                continue
            ids = torch.arange(1, counts[j] + 1, dtype=torch.int64, device=label_maps[j].device)
            t = ops.label_histogram(label_maps[i].contiguous(), label_maps[j].contiguous(), ids, counts[i])
            pairs.append((i, j, tuple(t.shape)))
            flat.append(t.reshape(-1))
    if not pairs:
        return {}
    host = torch.cat(flat).cpu().numpy()
    tables, at = {}, 0
    for i, j, shape in pairs:
        n = shape[0] * shape[1]
        tables[(i, j)] = host[at:at + n].reshape(shape)
        at += n
    return tables


def find_consensus_auto_association(frames, keyframe_selection: Literal['last', 'middle', 'score', 'first'] = 'last',
                                    *, network, store, config: Dict):
    """frames: FrameInfo-like objects (image 3*H*W, mask H*W index mask, segments_info, ti)
    -> (keyframe time index, consensus index mask H*W, merged ObjectInfo list)"""
    time_indices = [f.ti for f in frames]
    images, masks, pads = [], [], None
    for f in frames:
        image, pads = pad_divide_by(f.image, 16)
        mask, _ = pad_divide_by(f.mask, 16)
        images.append(image)
        masks.append(mask)

    # window-wide segment ids 1..total (ids of different frames must not collide); one-hot masks
    total = 0
    infos: Dict[int, ObjectInfo] = {}
    frame_segments: List[List[ObjectInfo]] = []
    one_hot: List[torch.Tensor] = []
    for i, f in enumerate(frames):
        segs, planes = [], []
        for seg in f.segments_info:
            total += 1
            new = ObjectInfo(total)
            new.copy_meta_info(seg)
            infos[total] = new
            segs.append(new)
            planes.append(masks[i] == seg.id)
        frame_segments.append(segs)
        one_hot.append(torch.stack(planes, dim=0).float() if planes else None)

    if keyframe_selection == 'last':
        key_i = len(frames) - 1
    elif keyframe_selection == 'first':
        key_i = 0
    elif keyframe_selection == 'middle':
        key_i = (len(frames) + 1) // 2
    else:
        raise NotImplementedError
    key_ti, key_image, key_mask = time_indices[key_i], images[key_i], one_hot[key_i]

    if total == 0:  # no detection in the whole window
        return key_ti, torch.zeros_like(frames[0].mask), []

    # ---- project every frame onto the keyframe; local label maps (0 = background, c+1 = c-th segment)
    label_maps: List[torch.Tensor] = []
    for i, (ti, image, planes) in enumerate(zip(time_indices, images, one_hot)):
        if planes is None:
            label_maps.append(None)
            continue
        if ti == key_ti:
            projected = torch.cat([torch.full_like(planes[0:1], 0.5), planes], dim=0)
        else:
            projected = spatial_alignment(ti, image, planes, key_ti, key_image, network, store, config)[0]
        label_maps.append(unpad(ops.index_mask(projected.contiguous()), pads))  # argmax per pixel, cropped
    counts = [len(s) for s in frame_segments]

    # ---- all intersections / areas in one go
    tables = pairwise_tables(label_maps, counts)
    areas: Dict[int, int] = {}
    local: Dict[int, Tuple[int, int]] = {}  # window-wide id -> (frame, local label)
    for i, segs in enumerate(frame_segments):
        if label_maps[i] is None:
            continue
        table = next((t for (a, b), t in tables.items() if a == i), None)
        if table is not None:
            per_label = table.sum(axis=1)
        else:
            other = next((t for (a, b), t in tables.items() if b == i), None)
            if other is not None:
                per_label = np.concatenate([[0], other.sum(axis=0)[:-1]])
            else:  # the only frame with detections: histogram against itself
                ids = torch.arange(1, counts[i] + 1, dtype=torch.int64, device=label_maps[i].device)
                per_label = ops.label_histogram(label_maps[i].contiguous(), label_maps[i].contiguous(), ids,
                                                counts[i]).cpu().numpy().sum(axis=1)
        for c, seg in enumerate(segs):
            areas[seg.id] = int(per_label[c + 1])
            local[seg.id] = (i, c + 1)

    # ---- greedy IoU > 0.5 matching, thing / stuff / untyped separately (consensus_automatic.py:183-223)
    matching = defaultdict(list)
    pairwise_iou = np.zeros((total, total), dtype=np.float32)
    for (i, j), table in tables.items():
        for status in (None, False, True):
            taken = set()
            for a in frame_segments[i]:
                if a.isthing != status:
                    continue
                for b in frame_segments[j]:
                    if b.isthing != status or b.id in taken:
                        continue
                    inter = int(table[local[a.id][1], local[b.id][1] - 1])
                    if inter == 0:
                        continue
                    iou = inter / (areas[a.id] + areas[b.id] - inter)
                    if iou > 0.5:
                        matching[a.id].append(b.id)
                        matching[b.id].append(a.id)
                        taken.add(b.id)
                        pairwise_iou[a.id - 1, b.id - 1] = iou
                        break
    pairwise_iou = pairwise_iou + pairwise_iou.T
    indicator = pairwise_iou > 0.49
    pairwise_iou = pairwise_iou * indicator

    selected = solve(pairwise_iou, indicator, total)

    # ---- paint the selected segments, large areas first, and merge the meta data of their matches
    output_mask = torch.zeros_like(frames[0].mask)
    output_info, chosen_area = [], {}
    for idx, on in enumerate(selected):
        if not on:
            continue
        oid = idx + 1
        chosen_area[oid] = areas[oid]
        info = infos[oid]
        for other in matching[oid]:
            info.merge(infos[other])
        output_info.append(info)
    for oid, _ in sorted(chosen_area.items(), key=lambda kv: kv[1], reverse=True):
        frame_i, label = local[oid]
        output_mask[label_maps[frame_i] == label] = oid
    return key_ti, output_mask, output_info
