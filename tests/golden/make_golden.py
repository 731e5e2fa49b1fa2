"""Generate the golden fixtures by running the REFERENCE itself (PyTorch CPU, fp32).

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
It imports the reference with the two shims of SURVEY.md §8c (stub `pulp`; force
`pretrained=False` for the ResNets), loads the recipe weights of workload/weights.py and writes
small .pt/.npz fixtures next to this file.  Nothing here is imported by the product or the tests.
"""
import hashlib
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

REF = os.environ.get('DEVA_REFERENCE_ROOT', '/root/reference')
sys.modules.setdefault('pulp', types.ModuleType('pulp'))
sys.path.insert(0, REF)
import deva.model.resnet as _R  # noqa: E402

_r18, _r50 = _R.resnet18, _R.resnet50
_R.resnet18 = lambda pretrained=True, extra_dim=0: _r18(pretrained=False, extra_dim=extra_dim)
_R.resnet50 = lambda pretrained=True, extra_dim=0: _r50(pretrained=False, extra_dim=extra_dim)
from deva.model.network import DEVA  # noqa: E402
from deva.inference.inference_core import DEVAInferenceCore  # noqa: E402
from deva.model import memory_utils as MU  # noqa: E402
from deva.inference.memory_manager import MemoryManager  # noqa: E402
from deva.utils.tensor_utils import pad_divide_by  # noqa: E402

from workload import synth, weights  # noqa: E402
import scenarios  # noqa: E402

warnings.filterwarnings('ignore')
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def build_reference(cfg, recipe='default'):
    net = DEVA(cfg).eval()
    spec = [(k, tuple(v.shape), v.dtype) for k, v in net.state_dict().items()]
    sd = weights.make_state_dict(spec, seed=0, recipe=recipe)
    net.load_weights(sd)
    return net, spec, sd


def gen_spec(spec, sd):
    h = hashlib.sha256()
    for k, _, _ in spec:
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    out = dict(sha256_seed0=h.hexdigest(),
               tensors=[[k, list(s), str(d).replace('torch.', '')] for k, s, d in spec])
    with open(os.path.join(HERE, 'state_dict_spec.json'), 'w') as f:
        json.dump(out, f, indent=0)


def gen_memory_ops():
    """get_similarity / do_softmax(top_k, usage) / readout  (memory_utils.py:6-76,
    memory_manager.py:64-75) plus the no-top-k softmax used by consolidation."""
    cases = {}
    for name, (n, hw, scale, seed) in {
            'small_peaky': (300, 48, 1.0, 11),
            'mid_flat': (1000, 100, 0.15, 12),     # flat similarities -> many near ties
            'ragged': (77, 33, 0.5, 13),            # sizes that are not multiples of any tile
    }.items():
        mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=seed, key_scale=scale)
        sim = MU.get_similarity(mk, ms, qk, qe, add_batch_dim=True)
        vals, idx = torch.topk(sim, k=30, dim=1)
        aff, usage = MU.do_softmax(sim.clone(), top_k=30, inplace=True, return_usage=True)
        v = synth.value_inputs(2, 512, n, seed=seed)
        mm = MemoryManager(synth.base_config())
        ro = mm._readout(aff[0], v)
        full = MU.do_softmax(sim)  # consolidation path
        cases[name] = dict(n=n, hw=hw, scale=scale, seed=seed, sim=sim[0].clone(),
                           topk_values=vals[0], topk_indices=idx[0].int(),
                           usage=usage[0], readout=ro, full_softmax=full[0])
    torch.save(cases, os.path.join(HERE, 'memory_ops.pt'))


def gen_stages(net):
    """Teacher-forced stage outputs at 96x128, 2 objects."""
    H, W, no = 96, 128, 2
    img = synth.FrameStream(H, W, seed=5).next().unsqueeze(0)
    ms, feat = net.encode_image(img)
    key, shr, sel = net.transform_key(feat)
    masks, sensory, readout = synth.stage_inputs(H, W, no)
    value, sens_deep = net.encode_mask(img, ms, sensory, masks, is_deep_update=True)
    sens_seg, logits, prob = net.segment(ms, readout, sensory, masks)
    out = dict(f16=ms[0], f8=ms[1], f4=ms[2], feat=feat, key=key, shrinkage=shr, selection=sel,
               value=value, sensory_deep=sens_deep, sensory_seg=sens_seg, logits=logits, prob=prob)
    torch.save({k: v.clone() for k, v in out.items()}, os.path.join(HERE, 'stages_96x128.pt'))


def gen_e2e(net):
    for name, sc in scenarios.E2E.items():
        def make_core(cfg):
            return DEVAInferenceCore(net, cfg)

        outs, core = scenarios.run_scenario(make_core, sc)
        mem = core.memory
        sizes = dict(work={b: mem.work_mem.size(b) for b in mem.work_mem.buckets},
                     long=({b: mem.long_mem.size(b) for b in mem.long_mem.buckets}
                           if mem.use_long_term else {}))
        # per-frame arrays: the channel count changes mid-clip when a second annotation arrives
        np.savez_compressed(
            os.path.join(HERE, f'e2e_{name}.npz'),
            **{f'prob_sub_{t}': p[:, ::2, ::2].numpy() for t, p in enumerate(outs)},
            argmax=np.stack([p.argmax(0).numpy().astype(np.uint8) for p in outs]),
            nchan=np.array([p.shape[0] for p in outs]),
            sizes=json.dumps(sizes))
        print(name, 'frames', len(outs), 'sizes', sizes)


def gen_peaky():
    """the second weight recipe (workload/weights.py:RECIPES['peaky']): a plain propagation clip and the
    tracker-consistent detection clip (workload/detections.py; matches, new buckets, purges, consolidation),
    both run by the reference itself"""
    from deva.inference.object_info import ObjectInfo
    from workload import detections
    net, _, _ = build_reference(synth.base_config(), recipe='peaky')
    for name, sc in scenarios.E2E_PEAKY.items():
        outs, core = scenarios.run_scenario(lambda cfg: DEVAInferenceCore(net, cfg), sc)
        mem = core.memory
        sizes = dict(work={b: mem.work_mem.size(b) for b in mem.work_mem.buckets},
                     long={b: mem.long_mem.size(b) for b in mem.long_mem.buckets})
        np.savez_compressed(os.path.join(HERE, f'e2e_{name}.npz'),
                            **{f'prob_sub_{t}': p[:, ::2, ::2].numpy() for t, p in enumerate(outs)},
                            argmax=np.stack([p.argmax(0).numpy().astype(np.uint8) for p in outs]),
                            nchan=np.array([p.shape[0] for p in outs]), sizes=json.dumps(sizes))
        print(name, 'frames', len(outs), 'sizes', sizes)
    # the detection clip runs on the DEFAULT recipe: it consolidates (usage-ranked prototypes), and with the peaky
    # recipe most usage counters underflow to exactly 0 -- ties that torch.topk breaks in an unspecified order
    net, _, _ = build_reference(synth.base_config())
    sc = scenarios.CONSISTENT
    holder = {}

    def make_core(cfg):
        holder['core'] = DEVAInferenceCore(net, cfg)
        return holder['core']

    outs, core, recorded = scenarios.run_consistent_detection_scenario(
        make_core, ObjectInfo, sc,
        record=lambda det, rec, frame_of: detections.record_on_package(holder['core'], det, ObjectInfo, rec, frame_of))
    mem = core.memory
    sizes = dict(work={b: mem.work_mem.size(b) for b in mem.work_mem.buckets},
                 long={b: mem.long_mem.size(b) for b in mem.long_mem.buckets})
    np.savez_compressed(os.path.join(HERE, 'e2e_consistent_detections.npz'),
                        **{f'prob_sub_{t}': p[:, ::2, ::2].numpy() for t, p in enumerate(outs)},
                        **{f'det_mask_{t}': m.numpy().astype(np.int32) for t, (m, _) in recorded.items()},
                        det_info=json.dumps({str(t): info for t, (_, info) in recorded.items()}),
                        nchan=np.array([p.shape[0] for p in outs]), sizes=json.dumps(sizes),
                        state=json.dumps(_manager_state(core.object_manager)))
    print('consistent detections: channels per frame', [p.shape[0] for p in outs], _manager_state(core.object_manager),
          sizes)


def gen_vos_example(net):
    """BASELINE config 1: example/vos bmx-trees, 4 frames 854x480, 2 objects, default flags.
    Frames are stored decoded (uint8) so the GPU box needs neither the reference nor a JPEG codec
    match; normalisation = ImageNet mean/std as deva/dataset/utils.py:8."""
    from PIL import Image
    base = os.path.join(REF, 'example', 'vos')
    names = ['00000', '00001', '00002', '00003']
    frames = np.stack([np.array(Image.open(os.path.join(base, 'JPEGImages', 'bmx-trees', n + '.jpg')).convert('RGB'))
                       for n in names])
    ann = np.array(Image.open(os.path.join(base, 'Annotations', 'bmx-trees', '00000.png')))
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    cfg = synth.base_config(enable_long_term_count_usage=False)
    core = DEVAInferenceCore(net, cfg)
    labels = [int(x) for x in np.unique(ann) if x != 0]
    outs = []
    for t in range(len(names)):
        img = (torch.from_numpy(frames[t]).permute(2, 0, 1).float() / 255 - mean) / std
        if t == 0:
            p = core.step(img, torch.from_numpy(ann.astype(np.int64)), labels, end=False)
        else:
            p = core.step(img, end=(t == len(names) - 1))
        outs.append(p)
    np.savez_compressed(os.path.join(HERE, 'e2e_vos_example.npz'), frames=frames, annotation=ann,
                        labels=np.array(labels),
                        prob_sub=np.stack([p[:, ::4, ::4].numpy() for p in outs]).astype(np.float32),
                        argmax=np.stack([p.argmax(0).numpy().astype(np.uint8) for p in outs]))
    print('vos example labels', labels, [tuple(p.shape) for p in outs])


def _manager_state(om):
    return dict(ids=[int(o.id) for o in om.obj_to_tmp_id], tmp=[int(t) for t in om.obj_to_tmp_id.values()],
                poke=[int(o.poke_count) for o in om.obj_to_tmp_id],
                cats=[[None if c is None else int(c) for c in o.category_ids] for o in om.obj_to_tmp_id],
                isthing=[o.isthing for o in om.obj_to_tmp_id])


def gen_merge():
    """match_and_merge (segment_merging.py:89-143) stand-alone, normal and incremental mode"""
    from deva.inference.object_info import ObjectInfo
    from deva.inference.object_manager import ObjectManager
    from deva.inference.segment_merging import match_and_merge
    out = {}
    for seed in (0, 1):
        for incremental in (False, True):
            ours, our_info, news, new_info = scenarios.merge_case(seed)
            om = ObjectManager()
            om.add_new_objects([ObjectInfo(**i) for i in our_info])
            merged = match_and_merge(ours, news, om, [ObjectInfo(**i) for i in new_info],
                                     incremental_mode=incremental)
            out[f'seed{seed}_inc{int(incremental)}'] = dict(onehot=merged.to(torch.uint8), state=_manager_state(om))
    torch.save(out, os.path.join(HERE, 'merge_cases.pt'))


def gen_detection_e2e(net):
    from deva.inference.object_info import ObjectInfo
    sc = scenarios.DETECTION
    outs, core = scenarios.run_detection_scenario(lambda cfg: DEVAInferenceCore(net, cfg), ObjectInfo, sc)
    np.savez_compressed(os.path.join(HERE, 'e2e_detections.npz'),
                        **{f'prob_sub_{t}': p[:, ::2, ::2].numpy() for t, p in enumerate(outs)},
                        nchan=np.array([p.shape[0] for p in outs]),
                        state=json.dumps(_manager_state(core.object_manager)))
    print('detections: channels per frame', [p.shape[0] for p in outs], _manager_state(core.object_manager)['ids'])


def gen_alignment(net):
    """spatial_alignment (consensus_associated.py:16-69): project a 2-object segmentation of one
    frame onto the next frame of the voting window; plus the keyframe projection of
    find_consensus_with_established_association (:82-160) on three frames"""
    from deva.inference.consensus_associated import find_consensus_with_established_association, spatial_alignment
    from deva.inference.image_feature_store import ImageFeatureStore
    cfg = synth.base_config()
    sc = scenarios.ALIGNMENT
    frames, masks = scenarios.alignment_inputs(sc)
    store = ImageFeatureStore(net, no_warning=True)
    out = spatial_alignment(0, frames[0], masks[0], 1, frames[1], net, store, cfg)
    store2 = ImageFeatureStore(net, no_warning=True)
    key_ti, consensus = find_consensus_with_established_association(
        [0, 1, 2], [f.clone() for f in frames], [m.clone() for m in masks], net, store2, cfg)
    torch.save(dict(aligned=out.clone(), keyframe=int(key_ti), consensus=consensus.clone()),
               os.path.join(HERE, 'alignment.pt'))


def gen_consensus_auto(net):
    """find_consensus_auto_association (consensus_automatic.py:82-290) on a 3-frame window.  The 0/1
    programme is solved by Gurobi / PuLP in the reference; neither exists here, so the reference's
    `solve_with_pulp` is replaced by the exact enumeration of the package (the optimum is unique on this
    window); everything else -- projection, IoU table, greedy matching, merging, painting -- is the
    reference's own code."""
    import importlib
    sys.path.insert(0, os.path.join(ROOT, 'tracking-anything-with-deva_amd', 'deva', 'inference'))
    spec = importlib.util.spec_from_file_location(
        '_pkg_consensus', os.path.join(ROOT, 'tracking-anything-with-deva_amd', 'deva', 'inference', 'consensus_automatic.py'))
    from deva.inference import consensus_automatic as CA
    from deva.inference.image_feature_store import ImageFeatureStore
    from deva.inference.object_info import ObjectInfo
    src = open(spec.origin).read()
    ns = {}
    exec(src[src.index('def solve_exact'):src.index('def solve(')], {'np': np, 'List': list, 'Tuple': tuple}, ns)
    captured = {}

    def solver(iou, ind, n):
        captured['iou'] = iou.copy()
        return ns['solve_exact'](iou, ind, n)

    CA.solve_with_pulp = solver
    CA.use_gurobi = False
    out = {}
    real_alignment = CA.spatial_alignment
    # 'network': the real projection (with recipe weights it is noise: no pair reaches IoU 0.5, which pins the
    # plumbing and the empty-table path); 'static': the projection replaced by "the scene does not move"
    # (scenarios.static_projection), which gives the table / matching / selection / painting real work
    for case, sel in (('network_last', 'last'), ('static_first', 'first'), ('static_middle', 'middle')):
        CA.spatial_alignment = real_alignment if case.startswith('network') else scenarios.static_projection
        frames = scenarios.consensus_inputs(scenarios.CONSENSUS, lambda **kw: ObjectInfo(**kw))
        store = ImageFeatureStore(net, no_warning=True)
        ti, mask, info = CA.find_consensus_auto_association(frames, keyframe_selection=sel, network=net, store=store,
                                                            config=synth.base_config())
        out[case] = dict(ti=int(ti), mask=mask.clone(), iou=torch.from_numpy(captured['iou']),
                         info=[dict(id=int(o.id), cats=list(o.category_ids), isthing=o.isthing, scores=list(o.scores))
                               for o in info])
    CA.spatial_alignment = real_alignment
    torch.save(out, os.path.join(HERE, 'consensus_auto.pt'))


def gen_driver_semionline(net):
    """the semi-online loop of evaluation/eval_with_detections.py:150-297 (restated against the public interface in
    tests/driver_loops.py) driven on the REFERENCE: buffering, vote_in_temporary_buffer -> find_consensus_auto_association
    (0/1 programme by the package's exact enumeration, like gen_consensus_auto: no PuLP / Gurobi here),
    incorporate_detection, propagation, clear_buffer.  Stored: the index masks the driver would write."""
    import importlib
    import driver_loops
    from deva.inference import consensus_automatic as CA
    from deva.inference.object_info import ObjectInfo
    spec = importlib.util.spec_from_file_location(
        '_pkg_consensus', os.path.join(ROOT, 'tracking-anything-with-deva_amd', 'deva', 'inference', 'consensus_automatic.py'))
    src = open(spec.origin).read()
    ns = {}
    exec(src[src.index('def solve_exact'):src.index('def solve(')], {'np': np, 'List': list, 'Tuple': tuple}, ns)
    CA.solve_with_pulp = lambda iou, ind, n: ns['solve_exact'](iou, ind, n)
    CA.use_gurobi = False
    frames, dets = driver_loops.semionline_clip()
    cfg = dict(synth.base_config(mem_every=2, max_missed_detection_count=2, max_num_objects=-1), num_voting_frames=3)
    masks, alive = driver_loops.semionline_loop(lambda c: DEVAInferenceCore(net, c), cfg, frames, dets, lambda **kw: ObjectInfo(**kw),
                                                num_voting_frames=3, detection_every=5, device='cpu')
    names = sorted(masks)
    np.savez_compressed(os.path.join(HERE, 'driver_semionline.npz'), names=np.array(names),
                        masks=np.stack([masks[n].numpy() for n in names]).astype(np.int32), alive=np.array(alive),
                        config=json.dumps({k: v for k, v in cfg.items() if isinstance(v, (int, float, bool, str))}))
    print('driver_semionline', len(names), 'frames, objects alive at the end', alive,
          'labels per frame', [sorted(set(masks[n].flatten().tolist())) for n in names])


def gen_read_memory(net):
    """DEVA.read_memory (network.py:72-92): dense full-softmax read, B=2, 2 objects, T=3 memory frames"""
    g = torch.Generator().manual_seed(31)
    B, no, T, h, w = 2, 2, 3, 5, 7
    args = dict(query_key=torch.randn(B, 64, h, w, generator=g), query_selection=torch.rand(B, 64, h, w, generator=g),
                memory_key=torch.randn(B, 64, T, h, w, generator=g), memory_shrinkage=torch.rand(B, 1, T, h, w, generator=g) + 1,
                memory_value=torch.randn(B, no, 512, T, h, w, generator=g))
    out = net.read_memory(**args)
    torch.save(dict(args=args, out=out.clone()), os.path.join(HERE, 'read_memory.pt'))
    print('read_memory', tuple(out.shape), float(out.abs().max()))


def gen_edge(net):
    from deva.inference.object_info import ObjectInfo
    out = scenarios.run_edge_cases(lambda cfg: DEVAInferenceCore(net, cfg), make_info=ObjectInfo)
    torch.save(out, os.path.join(HERE, 'edge_cases.pt'))
    print({k: (tuple(v.shape), float(v.float().max())) for k, v in out.items()})


def gen_api_surface():
    """public methods (name -> parameter list with defaults) of the classes on the drop-in boundary,
    read off the reference with inspect"""
    import inspect
    from deva.inference.image_feature_store import ImageFeatureStore
    from deva.inference.kv_memory_store import KeyValueMemoryStore
    from deva.inference.object_info import ObjectInfo
    from deva.inference.object_manager import ObjectManager
    out = {}
    for cls in (DEVA, DEVAInferenceCore, MemoryManager, KeyValueMemoryStore, ObjectManager, ObjectInfo, ImageFeatureStore):
        methods = {}
        for name, fn in inspect.getmembers(cls, predicate=inspect.isfunction):
            if fn.__qualname__.split('.')[0] != cls.__name__:
                continue  # inherited from nn.Module / object
            if name.startswith('_') and name not in ('__init__', '_segment', '_add_memory'):
                continue
            sig = inspect.signature(fn)
            methods[name] = [[p.name, str(p.kind), None if p.default is inspect._empty else repr(p.default)]
                             for p in sig.parameters.values()]
        props = [n for n, v in inspect.getmembers(cls) if isinstance(v, property)]
        out[cls.__name__] = dict(methods=methods, properties=props)
    # command-line surface of the evaluation drivers (eval_args.py:7-43)
    from argparse import ArgumentParser
    from deva.inference.eval_args import add_common_eval_args
    parser = ArgumentParser()
    add_common_eval_args(parser)
    out['eval_args'] = {a.dest: dict(flags=a.option_strings, default=a.default, nargs=a.nargs,
                                     type=None if a.type is None else a.type.__name__,
                                     switch=type(a).__name__ == '_StoreTrueAction')
                        for a in parser._actions if a.dest != 'help'}
    with open(os.path.join(HERE, 'api_surface.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    only = os.environ.get('ONLY')
    if only == 'read_memory':
        net, _, _ = build_reference(synth.base_config())
        gen_read_memory(net)
        sys.exit(0)
    if only == 'edge':
        net, _, _ = build_reference(synth.base_config())
        gen_edge(net)
        sys.exit(0)
    if only == 'api':
        gen_api_surface()
        sys.exit(0)
    if only == 'drivers':
        net, _, _ = build_reference(synth.base_config())
        gen_driver_semionline(net)
        sys.exit(0)
    if only == 'consensus':
        net, _, _ = build_reference(synth.base_config())
        gen_consensus_auto(net)
        sys.exit(0)
    if only == 'peaky':
        gen_peaky()
        sys.exit(0)
    if only == 'alignment':
        net, _, _ = build_reference(synth.base_config())
        gen_alignment(net)
        sys.exit(0)
    cfg = synth.base_config()
    net, spec, sd = build_reference(cfg)
    gen_spec(spec, sd)
    gen_memory_ops()
    gen_stages(net)
    gen_e2e(net)
    gen_vos_example(net)
    gen_merge()
    gen_detection_e2e(net)
    gen_alignment(net)
    gen_consensus_auto(net)
    gen_driver_semionline(net)
    gen_edge(net)
    gen_read_memory(net)
    gen_api_surface()
    gen_peaky()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
