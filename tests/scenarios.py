"""End-to-end propagation scenarios shared by the golden generator (reference), the CPU tests
(oracle) and the GPU tests (HIP runtime).  A scenario drives any object with the
`DEVAInferenceCore.step(image, mask, objects, end=...)` call pattern of
evaluation/eval_vos.py:167."""
import os
import sys
from typing import Callable, Dict, List

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from workload import synth  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# name -> scenario.  `second` = (frame index, object id) of a later partial annotation that
# introduces a new object (creates a second memory bucket, kv_memory_store.py:82-89).
E2E: Dict[str, Dict] = {
    # long-term on; 3 consolidations, the 3rd preceded by a least-usage eviction
    'lt_evict': dict(H=96, W=128, nobj=2, frames=45, second=None,
                     cfg=dict(mem_every=2, max_long_term_elements=80, num_prototypes=32)),
    # a second annotated object arrives at frame 7 -> two buckets, both consolidate
    'two_buckets': dict(H=96, W=128, nobj=2, frames=30, second=(7, 5),
                        cfg=dict(mem_every=2, max_long_term_elements=300, num_prototypes=32)),
    # working memory only (BASELINE config 2 style), size needing pad on both axes
    'no_lt': dict(H=100, W=150, nobj=3, frames=12, second=None,
                  cfg=dict(enable_long_term=False, enable_long_term_count_usage=False, mem_every=3)),
    # five objects, default flags
    'five_obj': dict(H=90, W=130, nobj=5, frames=9, second=None, cfg=dict()),
}


def run_scenario(make_core: Callable[[Dict], object], sc: Dict, device: str = 'cpu',
                 on_frame: Callable = None, perturb: Callable = None) -> List[torch.Tensor]:
    """Returns the per-frame `step` outputs ([no+1,H,W] probabilities, on CPU)."""
    cfg = synth.base_config(**sc['cfg'])
    core = make_core(cfg)
    stream = synth.FrameStream(sc['H'], sc['W'], seed=1)
    mask0 = synth.box_mask(sc['H'], sc['W'], sc['nobj'])
    objs = list(range(1, sc['nobj'] + 1))
    outs = []
    for t in range(sc['frames']):
        img = stream.next()
        if perturb is not None:
            img = perturb(img)
        img = img.to(device)
        end = (t == sc['frames'] - 1)
        if t == 0:
            p = core.step(img, mask0.to(device), objs, end=end)
        elif sc['second'] is not None and t == sc['second'][0]:
            oid = sc['second'][1]
            m = torch.zeros(sc['H'], sc['W'], dtype=torch.long)
            m[sc['H'] // 2:, :sc['W'] // 4] = oid
            p = core.step(img, m.to(device), [oid], end=end)
        else:
            p = core.step(img, end=end)
        outs.append(p.detach().float().cpu())
        if on_frame is not None:
            on_frame(t, core)
    return outs, core


# ---------------------------------------------------------------------------------------------
# detections + propagation (evaluation/eval_with_detections.py:280-297, "online" setting): an
# image-level detection mask is merged every `every`-th frame, frames in between are propagated.
DETECTION = dict(H=96, W=128, frames=13, every=4,
                 cfg=dict(mem_every=2, max_missed_detection_count=1, max_num_objects=-1))


def detection_mask(sc, t):
    """index mask with ids 10 (drifting box), 20 (static box), 30 (only in the first detection),
    40 (appears from the second detection on); plus their category / thing flags"""
    H, W = sc['H'], sc['W']
    m = torch.zeros(H, W, dtype=torch.long)
    x = 8 + 2 * t
    m[10:50, x:x + 40] = 10
    m[55:90, 70:120] = 20
    info = [dict(id=10, category_id=3, isthing=True), dict(id=20, category_id=7, isthing=False)]
    if t == 0:
        m[60:90, 5:35] = 30
        info.append(dict(id=30, category_id=3, isthing=True))
    else:
        m[2:20, 90:125] = 40
        info.append(dict(id=40, category_id=None, isthing=None))
    return m, info


def run_detection_scenario(make_core, make_info, sc, device='cpu'):
    """make_info(id=..., category_id=..., isthing=...) builds the implementation's ObjectInfo"""
    import numpy as np
    np.random.seed(0)  # ObjectManager draws replacement ids from np.random on id collisions
    cfg = synth.base_config(**sc['cfg'])
    core = make_core(cfg)
    stream = synth.FrameStream(sc['H'], sc['W'], seed=1)
    outs = []
    for t in range(sc['frames']):
        img = stream.next().to(device)
        if t % sc['every'] == 0:
            m, info = detection_mask(sc, t)
            p = core.incorporate_detection(img, m.to(device), [make_info(**i) for i in info])
        else:
            p = core.step(img, end=(t == sc['frames'] - 1))
        outs.append(p.detach().float().cpu())
    return outs, core


# ---------------------------------------------------------------------------------------------
# the "peaky" weight recipe (workload/weights.py:RECIPES): sharper affinities and logits, so that the
# north-star criteria (1e-3 max-abs free-running, argmax-identical masks) are testable as written
E2E_PEAKY: Dict[str, Dict] = {
    'peaky': dict(H=96, W=128, nobj=3, frames=14, second=None, cfg=dict(mem_every=3)),
}

# tracker-consistent detections (workload/detections.py): re-detections that match (IoU 0.875), new
# segments that spawn objects in new buckets, objects that go unseen and are purged -- BASELINE
# configs[2]'s merge / purge / multi-object memory path.  Default recipe (it consolidates: with the peaky recipe most
# usage counters underflow to exactly 0, ties that torch.topk ranks in an unspecified order).
CONSISTENT = dict(H=96, W=128, frames=17, every=3, segments=4, new_per_frame=1,
                  cfg=dict(mem_every=2, max_missed_detection_count=1, max_num_objects=-1, max_mid_term_frames=6,
                           min_mid_term_frames=3, num_prototypes=32, max_long_term_elements=300))


def run_consistent_detection_scenario(make_core, make_info, sc, device='cpu', record=None, replay=None,
                                      perturb=None, on_frame=None):
    """Drives `incorporate_detection` every `every`-th frame and `step` in between.
    record(detector, recorded, frame_of) -> context manager (workload.detections.record_on_*) under which the
    run GENERATES its detections from its own forward masks (they end up in the returned dict);
    replay = {frame: (mask, info)} recorded by another run, fed through the public interface.
    -> (per-frame outputs on the CPU, core, {frame: (mask, info)})"""
    import numpy as np
    from workload.detections import ConsistentDetector
    np.random.seed(0)
    assert (record is None) != (replay is None)
    cfg = synth.base_config(**sc['cfg'])
    core = make_core(cfg)
    H, W = sc['H'], sc['W']
    stream = synth.FrameStream(H, W, seed=sc.get('seed', 1))
    detector = ConsistentDetector(H, W, sc['segments'], sc['new_per_frame'])
    recorded, now, outs = {}, [0], []
    for t in range(sc['frames']):
        now[0] = t
        img = stream.next()
        if perturb is not None:
            img = perturb(img)
        img = img.to(device)
        if t % sc['every'] == 0:
            if record is not None:
                with record(detector, recorded, lambda: now[0]):
                    p = core.incorporate_detection(img, torch.zeros(H, W, dtype=torch.long, device=device), [])
            else:
                m, info = replay[t]
                p = core.incorporate_detection(img, m.to(device), [make_info(**i) for i in info])
        else:
            p = core.step(img, end=(t == sc['frames'] - 1))
        outs.append(p.detach().float().cpu())
        if on_frame is not None:
            on_frame(t, core)
    return outs, core, (recorded if record is not None else replay)


def load_consistent_golden(golden_dir):
    """-> (npz, {frame: (mask, info)}) of tests/golden/e2e_consistent_detections.npz"""
    import json
    import numpy as np
    g = np.load(os.path.join(golden_dir, 'e2e_consistent_detections.npz'))
    info = json.loads(str(g['det_info']))
    return g, {int(t): (torch.from_numpy(g[f'det_mask_{t}'].astype('int64')), i) for t, i in info.items()}


def manager_state(om):
    """the observable object table of an ObjectManager (ids in tmp order, poke counters, category votes)"""
    return dict(ids=[int(o.id) for o in om.obj_to_tmp_id], tmp=[int(t) for t in om.obj_to_tmp_id.values()],
                poke=[int(o.poke_count) for o in om.obj_to_tmp_id],
                cats=[[None if c is None else int(c) for c in o.category_ids] for o in om.obj_to_tmp_id],
                isthing=[o.isthing for o in om.obj_to_tmp_id])


def merge_case(seed):
    """inputs of a stand-alone match_and_merge call: propagated tmp-id mask with 3 objects
    (thing / stuff / untyped), detections that match, overlap too little, are new, or overlap an
    object of another type"""
    g = torch.Generator().manual_seed(seed)
    H, W = 64, 80
    ours = torch.zeros(H, W, dtype=torch.long)
    ours[5:30, 5:35] = 1
    ours[35:60, 10:45] = 2
    ours[10:40, 50:75] = 3
    our_info = [dict(id=11, category_id=1, isthing=True), dict(id=12, category_id=2, isthing=False),
                dict(id=13, category_id=None, isthing=None)]
    news = torch.zeros(H, W, dtype=torch.long)
    dx = int(torch.randint(0, 4, (1,), generator=g))
    news[6:30, 6 + dx:36 + dx] = 101    # thing, IoU > 0.5 with object 11
    news[50:64, 10:30] = 102            # stuff, small overlap with object 12 -> new object
    news[12:40, 52:76] = 103            # thing over the untyped object 13 -> no cross-type match, new
    news[0:4, 60:80] = 104              # untyped, new
    new_info = [dict(id=101, category_id=1, isthing=True), dict(id=102, category_id=2, isthing=False),
                dict(id=103, category_id=5, isthing=True), dict(id=104, category_id=None, isthing=None)]
    return ours, our_info, news, new_info


# ---------------------------------------------------------------------------------------------
# semi-online voting window: spatial alignment of per-frame segmentations onto a keyframe
# (deva/inference/consensus_associated.py)
ALIGNMENT = dict(H=96, W=128, nobj=2, frames=3)


def alignment_inputs(sc):
    """three coherent frames and a soft 2-object segmentation for each (boxes that drift a little)"""
    stream = synth.FrameStream(sc['H'], sc['W'], seed=8)
    frames = [stream.next() for _ in range(sc['frames'])]
    masks = []
    for t in range(sc['frames']):
        m = torch.zeros(sc['nobj'], sc['H'], sc['W'])
        m[0, 10 + 2 * t:50 + 2 * t, 12 + 3 * t:60 + 3 * t] = 0.9 - 0.1 * t
        m[1, 40:90, 70 - 2 * t:120 - 2 * t] = 0.8
        masks.append(m)
    return frames, masks


# ---------------------------------------------------------------------------------------------
# API edge paths of DEVAInferenceCore (inference_core.py:55-113,137-290): warnings instead of
# exceptions, soft-mask annotation, feature-cache control, detection rounds without segments
EDGE = dict(H=80, W=112)


def run_edge_cases(make_core, device='cpu', make_info=None):
    """Returns {name: tensor or list} of everything observable; identical calls are made on the
    reference (golden generator), on the package with emulated ops (CPU) and on the GPU.
    make_info(id=..., category_id=..., isthing=...) builds the implementation's ObjectInfo."""
    import warnings
    H, W = EDGE['H'], EDGE['W']
    stream = synth.FrameStream(H, W, seed=21)
    frames = [stream.next().to(device) for _ in range(6)]
    out = {}

    # 1. propagating before anything was annotated: a RuntimeWarning and an all-zero 1*H*W map
    core = make_core(synth.base_config())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        p = core.step(frames[0])
    out['no_memory_prob'] = p.detach().float().cpu()
    out['no_memory_warned'] = torch.tensor(float(any(issubclass(x.category, RuntimeWarning) for x in w)))

    # 2. soft (probability) masks, objects implied by the channel order; last frame with end=True
    core = make_core(synth.base_config(mem_every=2))
    soft = torch.zeros(2, H, W)
    soft[0, 8:40, 10:60] = 0.85
    soft[1, 30:70, 50:100] = 0.7
    seq = [core.step(frames[0], soft.to(device), hard_mask=False)]
    seq += [core.step(frames[t]) for t in (1, 2, 3)]
    seq.append(core.step(frames[4], end=True))
    out['soft_seq'] = torch.stack([s.detach().float().cpu() for s in seq])
    out['soft_ids'] = torch.tensor(core.object_manager.all_obj_ids)
    out['soft_work_size'] = torch.tensor(core.memory.work_mem.size(0))

    # 3. feature-cache control: an overridden cache index survives the step when asked to
    core = make_core(synth.base_config())
    m = synth.box_mask(H, W, 2).to(device)
    core.step(frames[0], m, [1, 2], image_ti_override=77, delete_buffer=False)
    out['cache_len_kept'] = torch.tensor(len(core.image_feature_store))
    core.image_feature_store.delete(77)
    out['cache_len_after_delete'] = torch.tensor(len(core.image_feature_store))
    core.step(frames[1])
    out['cache_len_default'] = torch.tensor(len(core.image_feature_store))

    # 4. a detection round without any segment on an empty tracker: "Empty object mask!" and a
    # background-only answer; the tracker stays empty
    core = make_core(synth.base_config(max_missed_detection_count=1, max_num_objects=-1))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        p = core.incorporate_detection(frames[0], torch.zeros(H, W, dtype=torch.long, device=device), [])
    out['empty_detection_prob'] = p.detach().float().cpu()
    out['empty_detection_warned'] = torch.tensor(float(any('Empty object mask' in str(x.message) for x in w)))
    out['empty_detection_objects'] = torch.tensor(core.object_manager.num_obj)

    # 5. every tracked object disappears: two detected objects, then detection rounds that see nothing
    # (max_missed_detection_count=1) until the tracker and its memories are empty; propagation
    # afterwards is the "no memory" case again, and a new detection starts over
    if make_info is not None:
        core = make_core(synth.base_config(max_missed_detection_count=1, max_num_objects=-1, mem_every=2))
        det = torch.zeros(H, W, dtype=torch.long)
        det[10:40, 10:50] = 5
        det[45:75, 60:105] = 9
        info = lambda: [make_info(id=5, category_id=1, isthing=True), make_info(id=9, category_id=2, isthing=True)]
        nothing = torch.zeros(H, W, dtype=torch.long, device=device)
        seq, counts = [], []
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            seq.append(core.incorporate_detection(frames[0], det.to(device), info()))
            counts.append(core.object_manager.num_obj)
            seq.append(core.step(frames[1]))
            for t in (2, 3, 4):
                seq.append(core.incorporate_detection(frames[t], nothing, []))
                counts.append(core.object_manager.num_obj)
            engaged = core.memory.engaged
            seq.append(core.step(frames[5]))
            seq.append(core.incorporate_detection(frames[5], det.to(device), info()))
            counts.append(core.object_manager.num_obj)
        for i, p in enumerate(seq):
            out[f'vanish_{i}'] = p.detach().float().cpu()
        out['vanish_counts'] = torch.tensor(counts)
        out['vanish_engaged_after_purge'] = torch.tensor(float(engaged))
    return out


# ---------------------------------------------------------------------------------------------
# semi-online consensus with inferred association (deva/inference/consensus_automatic.py): a window of
# three frames with per-frame detections whose ids are unrelated between frames
CONSENSUS = dict(H=96, W=128, frames=3)


class WindowFrame:
    """the fields of deva/inference/frame_utils.py:FrameInfo the consensus uses"""

    def __init__(self, image, mask, segments_info, ti):
        self.image, self.mask, self.segments_info, self.ti = image, mask, segments_info, ti


def consensus_inputs(sc, make_info):
    """frame t: a thing box drifting right (always detected), a stuff box (missed in frame 1), an untyped
    box (only in frame 2), and a spurious thing detection in frame 0 only; ids differ from frame to frame"""
    stream = synth.FrameStream(sc['H'], sc['W'], seed=12)
    out = []
    for t in range(sc['frames']):
        img = stream.next()
        m = torch.zeros(sc['H'], sc['W'], dtype=torch.long)
        info = []
        m[12:52, 10 + 2 * t:58 + 2 * t] = 7 + 10 * t
        info.append(make_info(id=7 + 10 * t, category_id=2, isthing=True, score=0.9 - 0.1 * t))
        if t != 1:
            m[56:92, 66:122] = 3 + t
            info.append(make_info(id=3 + t, category_id=5, isthing=False, score=0.7))
        if t == 2:
            m[2:10, 100:126] = 99
            info.append(make_info(id=99, category_id=None, isthing=None, score=None))
        if t == 0:
            m[60:90, 4:30] = 50
            info.append(make_info(id=50, category_id=2, isthing=True, score=0.4))
        out.append(WindowFrame(img, m, info, ti=10 + t))
    return out


def static_projection(src_ti, src_image, src_mask, tar_ti, tar_image, network, store, config):
    """stand-in for spatial_alignment in the consensus tests: 'the scene does not move' -- the source
    segmentation with a 0.5 background plane, batch dimension in front (recipe weights project noise)"""
    return torch.cat([torch.full_like(src_mask[0:1], 0.5), src_mask], dim=0).unsqueeze(0)


def run_consensus_cases(network, make_store, make_info, device='cpu'):
    """the three cases of tests/golden/consensus_auto.pt on the package under test"""
    import deva.inference.consensus_automatic as CA
    from workload import synth as _synth
    real = CA.spatial_alignment
    out = {}
    try:
        for case, sel in (('network_last', 'last'), ('static_first', 'first'), ('static_middle', 'middle')):
            CA.spatial_alignment = real if case.startswith('network') else static_projection
            frames = consensus_inputs(CONSENSUS, make_info)
            for f in frames:
                f.image, f.mask = f.image.to(device), f.mask.to(device)
            ti, mask, info = CA.find_consensus_auto_association(frames, keyframe_selection=sel, network=network,
                                                                store=make_store(), config=_synth.base_config())
            out[case] = dict(ti=int(ti), mask=mask.cpu(),
                             info=[dict(id=int(o.id), cats=list(o.category_ids), isthing=o.isthing, scores=list(o.scores))
                                   for o in info])
    finally:
        CA.spatial_alignment = real
    return out


def check_consensus_cases(got, golden):
    for case, g in golden.items():
        assert got[case]['ti'] == g['ti'], case
        assert got[case]['info'] == g['info'], (case, got[case]['info'], g['info'])
        if case.startswith('static'):
            assert torch.equal(got[case]['mask'], g['mask']), case
        else:  # projected noise: only the decision (nothing reaches IoU 0.5) is pinned
            assert got[case]['mask'].unique().tolist() == g['mask'].unique().tolist(), case
