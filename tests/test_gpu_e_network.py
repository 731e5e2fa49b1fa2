"""The network stages and the full frame loop on the MI355X against (a) the golden vectors produced
by the reference itself and (b) the CPU oracle run side by side on the GPU box.

Tolerances (BASELINE.json north_star): soft outputs within 1e-3 max-abs of the reference CPU path; hard
masks argmax-identical at every pixel whose reference top-1/top-2 margin exceeds twice the BOUND of the
clip -- 1e-3, or 10x the reference's own drift under a 1e-6 input perturbation where that is larger and a
differing discrete decision has been explained (tests/memory_audit.py:Drift; the margin never depends on
the error under test).  With the default recipe many pixels sit at near-ties that the reference flips
against itself; the "peaky" recipe (workload/weights.py) has a noise floor of ~3e-4 and is held to the
north-star numbers as written (strict=True).  Teacher-forced per-stage checks use a 2e-4 relative bound."""
import json
import os

import numpy as np
import pytest
import torch

import memory_audit
import scenarios
from gpu_util import dev, max_err
from oracle import deva_oracle as O
from workload import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope='module')
def network(recipe_state_dict):
    from deva.model.network import DEVA
    sd, _ = recipe_state_dict
    net = DEVA(synth.base_config())
    net.load_weights(sd)
    return net.to(dev()).eval()


_margin_aware_mismatch = memory_audit.margin_aware_mismatch
_Drift = memory_audit.Drift


@pytest.fixture(scope='module')
def peaky_network(peaky_state_dict):
    from deva.model.network import DEVA
    net = DEVA(synth.base_config())
    net.load_weights(peaky_state_dict)
    return net.to(dev()).eval()


def _paired_clip(tag, stride, hip_frames, noisy_frames, hip_tap_marks, ref_tap_marks, hip_tap, ref_tap,
                 reference_outputs, strict=False):
    """frame-by-frame comparison of a HIP run against reference outputs, with the selection audit
    against the live oracle run (`*_marks[t]` = number of reads recorded up to and including frame t)"""
    drift = _Drift(tag, stride=stride, strict=strict)
    for t, p in enumerate(hip_frames):
        lo_h, hi_h = (hip_tap_marks[t - 1] if t else 0), hip_tap_marks[t]
        lo_r, hi_r = (ref_tap_marks[t - 1] if t else 0), ref_tap_marks[t]
        drift.audit_reads(t, hip_tap.reads[lo_h:hi_h], ref_tap.reads[lo_r:hi_r])
        drift.add(p, reference_outputs[t], None if noisy_frames is None else noisy_frames[t])
    drift.finish()
    return drift


def test_stages_teacher_forced(network, golden_dir):
    g = torch.load(os.path.join(golden_dir, 'stages_96x128.pt'))
    H, W, no = 96, 128, 2
    img = synth.FrameStream(H, W, seed=5).next().unsqueeze(0).to(dev())
    ms, feat = network.encode_image(img)
    key, shr, sel = network.transform_key(feat)
    masks, sensory, readout = (t.to(dev()) for t in synth.stage_inputs(H, W, no))
    value, sens_deep = network.encode_mask(img, ms, sensory, masks)
    sens_seg, logits, prob = network.segment(ms, readout, sensory, masks)
    torch.cuda.synchronize()
    got = dict(f16=ms[0], f8=ms[1], f4=ms[2], feat=feat, key=key, shrinkage=shr, selection=sel,
               value=value, sensory_deep=sens_deep, sensory_seg=sens_seg, logits=logits, prob=prob)
    report = {}
    for k, v in got.items():
        assert v.shape == g[k].shape, k
        report[k] = max_err(v, g[k]) / max(1.0, g[k].abs().max().item())
    print('stage rel errors:', json.dumps({k: float(f'{v:.3e}') for k, v in report.items()}))
    for k, v in report.items():
        assert v <= 2e-4, (k, v)
    assert max_err(prob, g['prob']) <= 1e-3 and max_err(logits, g['logits']) <= 1e-3


@pytest.mark.parametrize('name', list(scenarios.E2E))
def test_e2e_against_reference_golden(network, golden_dir, recipe_state_dict, name):
    from deva.inference.inference_core import DEVAInferenceCore
    P, _ = recipe_state_dict
    sc = scenarios.E2E[name]
    hip_marks, ref_marks = [], []
    with memory_audit.ReadTap() as hip_tap:
        outs, core = scenarios.run_scenario(lambda cfg: DEVAInferenceCore(network, cfg), sc, device=dev(),
                                            on_frame=lambda t, c: hip_marks.append(len(hip_tap.reads)))
    g = np.load(os.path.join(golden_dir, f'e2e_{name}.npz'))
    assert [p.shape[0] for p in outs] == g['nchan'].tolist()
    sizes = json.loads(str(g['sizes']))
    mem = core.memory
    assert {str(b): mem.work_mem.size(b) for b in mem.work_mem.buckets} == sizes['work']
    if mem.use_long_term:
        assert {str(b): mem.long_mem.size(b) for b in mem.long_mem.buckets} == sizes['long']
    # the live oracle on the same frames: its top-k selections are what the HIP run's are audited against
    with memory_audit.OracleTap() as ref_tap:
        scenarios.run_scenario(lambda cfg: O.OracleCore(P, cfg), sc,
                               on_frame=lambda t, c: ref_marks.append(len(ref_tap.reads)))
    # the reference's own sensitivity on this clip: oracle on frames perturbed by 1e-6 relative noise
    gen = torch.Generator().manual_seed(0)
    noisy_outs, _ = scenarios.run_scenario(lambda cfg: O.OracleCore(P, cfg), dict(sc),
                                           perturb=lambda img: img * (1 + 1e-6 * torch.randn(img.shape, generator=gen)))
    _paired_clip(name, 2, [p[:, ::2, ::2] for p in outs], [p[:, ::2, ::2] for p in noisy_outs], hip_marks, ref_marks,
                 hip_tap, ref_tap, [torch.from_numpy(g[f'prob_sub_{t}']) for t in range(len(outs))])


def test_e2e_peaky_against_reference_golden(peaky_network, golden_dir, peaky_state_dict):
    """the peaky recipe against the reference's own outputs, held to the north-star numbers as written:
    <= 1e-3 max-abs on every frame (no floor multiplier) and no argmax flip at a margin above 2e-3"""
    from deva.inference.inference_core import DEVAInferenceCore
    sc = scenarios.E2E_PEAKY['peaky']
    hip_marks, ref_marks = [], []
    with memory_audit.ReadTap() as hip_tap:
        outs, core = scenarios.run_scenario(lambda cfg: DEVAInferenceCore(peaky_network, cfg), sc, device=dev(),
                                            on_frame=lambda t, c: hip_marks.append(len(hip_tap.reads)))
    g = np.load(os.path.join(golden_dir, 'e2e_peaky.npz'))
    with memory_audit.OracleTap() as ref_tap:
        scenarios.run_scenario(lambda cfg: O.OracleCore(peaky_state_dict, cfg), sc,
                               on_frame=lambda t, c: ref_marks.append(len(ref_tap.reads)))
    gen = torch.Generator().manual_seed(0)
    noisy_outs, _ = scenarios.run_scenario(lambda cfg: O.OracleCore(peaky_state_dict, cfg), dict(sc),
                                           perturb=lambda img: img * (1 + 1e-6 * torch.randn(img.shape, generator=gen)))
    _paired_clip('peaky', 2, [p[:, ::2, ::2] for p in outs], [p[:, ::2, ::2] for p in noisy_outs], hip_marks, ref_marks,
                 hip_tap, ref_tap, [torch.from_numpy(g[f'prob_sub_{t}']) for t in range(len(outs))], strict=True)


def test_480p_five_objects_peaky_recipe_north_star_as_written(peaky_network, peaky_state_dict):
    """BASELINE configs[1] shape with the peaky recipe, free-running HIP vs the CPU oracle: <= 1e-3 max-abs on
    every frame and argmax-identical at every pixel with a reference margin above 2e-3 (strict)"""
    from deva.inference.inference_core import DEVAInferenceCore
    P = peaky_state_dict
    cfg = synth.base_config(enable_long_term=False, enable_long_term_count_usage=False)
    H, W, no, frames = 480, 854, 5, 7
    hip, orc, noisy = DEVAInferenceCore(peaky_network, cfg), O.OracleCore(P, cfg), O.OracleCore(P, cfg)
    stream = synth.FrameStream(H, W, seed=2)
    mask0 = synth.box_mask(H, W, no)
    objs = list(range(1, no + 1))
    gen = torch.Generator().manual_seed(0)
    drift = _Drift('480p/5obj/peaky', strict=True)
    for t in range(frames):
        img = stream.next()
        img_n = img * (1 + 1e-6 * torch.randn(img.shape, generator=gen))
        first = (mask0, objs) if t == 0 else (None, None)
        with memory_audit.ReadTap() as hip_tap:
            a = hip.step(img.to(dev()), None if first[0] is None else first[0].to(dev()), first[1])
        with memory_audit.OracleTap() as ref_tap:
            b = orc.step(img, first[0], first[1])
        c = noisy.step(img_n, first[0], first[1])
        drift.audit_reads(t, hip_tap.reads, ref_tap.reads)
        drift.add(a.cpu(), b, c)
    report = drift.finish()
    print('480p/5obj/peaky:', json.dumps({k: float(f'{v:.3g}') for k, v in report.items()}))


def test_consistent_detection_clip_against_reference_golden(peaky_network, golden_dir):
    """BASELINE configs[2]'s merge / purge / multi-bucket path on the HIP kernels: the reference's recorded
    tracker-consistent detections replayed through incorporate_detection (17 frames, 4 segments each):
    identical object table and bank sizes; soft outputs <= 1e-3 until the first merged hard mask that differs
    at a forward-argmax near-tie (a different hard mask is a different memory frame from there on)"""
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    g, golden_dets = scenarios.load_consistent_golden(golden_dir)
    sc = scenarios.CONSISTENT
    outs, core, _ = scenarios.run_consistent_detection_scenario(lambda cfg: DEVAInferenceCore(peaky_network, cfg),
                                                                ObjectInfo, sc, device=dev(), replay=golden_dets)
    assert [p.shape[0] for p in outs] == g['nchan'].tolist()
    assert scenarios.manager_state(core.object_manager) == json.loads(str(g['state']))
    sizes = json.loads(str(g['sizes']))
    mem = core.memory
    assert {str(b): mem.work_mem.size(b) for b in mem.work_mem.buckets} == sizes['work']
    assert {str(b): mem.long_mem.size(b) for b in mem.long_mem.buckets} == sizes['long']
    diverged = None
    for t, p in enumerate(outs):
        ref = torch.from_numpy(g[f'prob_sub_{t}'])
        if t % sc['every'] == 0:
            differ = int((p[:, ::2, ::2].argmax(0) != ref.argmax(0)).sum())
            print(f'frame {t} (detection): merged masks differ at {differ} of {ref[0].numel()} sampled pixels')
            assert differ <= 2e-3 * ref[0].numel() or diverged is not None, t
            if differ and diverged is None:
                diverged = t
        else:
            err = (p[:, ::2, ::2] - ref).abs().max().item()
            print(f'frame {t}: max-abs {err:.2e}' + ('' if diverged is None else f' (hard masks differ since frame {diverged})'))
            assert err <= 1e-3 or diverged is not None, (t, err)
            assert err <= 0.3, (t, err)
    assert diverged is None or diverged >= 3


def test_vos_example_against_reference_golden(network, golden_dir, recipe_state_dict):
    """BASELINE config 1: example/vos bmx-trees, 854x480 real frames, 2 objects, default flags"""
    from deva.inference.inference_core import DEVAInferenceCore
    P, _ = recipe_state_dict
    g = np.load(os.path.join(golden_dir, 'e2e_vos_example.npz'))
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    cfg = synth.base_config(enable_long_term_count_usage=False)
    core, clean, noisy = DEVAInferenceCore(network, cfg), O.OracleCore(P, cfg), O.OracleCore(P, cfg)
    labels = g['labels'].tolist()
    n = g['frames'].shape[0]
    ann = torch.from_numpy(g['annotation'].astype(np.int64))
    gen = torch.Generator().manual_seed(0)
    drift = _Drift('vos example', stride=4)
    for t in range(n):
        img = (torch.from_numpy(g['frames'][t]).permute(2, 0, 1).float() / 255 - mean) / std
        img_n = img * (1 + 1e-6 * torch.randn(img.shape, generator=gen))
        first, last = (ann, labels) if t == 0 else (None, None), (t == n - 1)
        with memory_audit.ReadTap() as hip_tap:
            p = core.step(img.to(dev()), None if first[0] is None else first[0].to(dev()), first[1], end=last)
        with memory_audit.OracleTap() as ref_tap:
            clean.step(img, first[0], first[1], end=last)
        pn = noisy.step(img_n, first[0], first[1], end=last)
        drift.audit_reads(t, hip_tap.reads, ref_tap.reads)
        ref = torch.from_numpy(g['prob_sub'][t])  # the reference's own output
        drift.add(p.cpu()[:, ::4, ::4], ref, pn[:, ::4, ::4])
    drift.finish()


def test_480p_five_objects_against_oracle(network, recipe_state_dict):
    """BASELINE config 2 shape (480x854 -> 480x864, 5 objects, working memory only), 7 frames,
    HIP runtime vs the CPU oracle on identical inputs."""
    from deva.inference.inference_core import DEVAInferenceCore
    P, _ = recipe_state_dict
    cfg = synth.base_config(enable_long_term=False, enable_long_term_count_usage=False)
    H, W, no, frames = 480, 854, 5, 7
    hip, orc, noisy = DEVAInferenceCore(network, cfg), O.OracleCore(P, cfg), O.OracleCore(P, cfg)
    stream = synth.FrameStream(H, W, seed=2)
    mask0 = synth.box_mask(H, W, no)
    objs = list(range(1, no + 1))
    gen = torch.Generator().manual_seed(0)
    drift = _Drift('480p/5obj')
    for t in range(frames):
        img = stream.next()
        img_n = img * (1 + 1e-6 * torch.randn(img.shape, generator=gen))
        first = (mask0, objs) if t == 0 else (None, None)
        with memory_audit.ReadTap() as hip_tap:
            a = hip.step(img.to(dev()), None if first[0] is None else first[0].to(dev()), first[1])
        with memory_audit.OracleTap() as ref_tap:
            b = orc.step(img, first[0], first[1])
        c = noisy.step(img_n, first[0], first[1])
        drift.audit_reads(t, hip_tap.reads, ref_tap.reads)
        drift.add(a.cpu(), b, c)
    drift.finish()


def test_480p_lockstep_teacher_forced(network, recipe_state_dict):
    """Every stage of every frame at full 480x864 size on IDENTICAL inputs (tests/lockstep.py)."""
    import lockstep
    P, _ = recipe_state_dict
    worst = lockstep.run(network, P, 480, 864, 2, 7, dev())
    print('lockstep 480p worst relative errors:', json.dumps({k: float(f'{v:.2e}') for k, v in worst.items()}))


def test_1080p_lockstep_teacher_forced(network, recipe_state_dict):
    """The same at BASELINE's 1080p size (1088x1920 padded, 8 160 queries), one object, 3 frames:
    the CPU oracle needs a few seconds per frame there, so the clip is short."""
    import lockstep
    P, _ = recipe_state_dict
    worst = lockstep.run(network, P, 1088, 1920, 1, 3, dev())
    print('lockstep 1080p worst relative errors:', json.dumps({k: float(f'{v:.2e}') for k, v in worst.items()}))


def test_detection_clip_against_reference_golden(network, golden_dir):
    """incorporate_detection (match_and_merge on the histogram / paint kernels) + propagation, online
    setting, against the reference's outputs and object-manager state"""
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    outs, core = scenarios.run_detection_scenario(lambda cfg: DEVAInferenceCore(network, cfg), ObjectInfo,
                                                  scenarios.DETECTION, device=dev())
    g = np.load(os.path.join(golden_dir, 'e2e_detections.npz'))
    assert [p.shape[0] for p in outs] == g['nchan'].tolist()
    om = core.object_manager
    state = dict(ids=[int(o.id) for o in om.obj_to_tmp_id], tmp=[int(t) for t in om.obj_to_tmp_id.values()],
                 poke=[int(o.poke_count) for o in om.obj_to_tmp_id],
                 cats=[[None if c is None else int(c) for c in o.category_ids] for o in om.obj_to_tmp_id],
                 isthing=[o.isthing for o in om.obj_to_tmp_id])
    assert state == json.loads(str(g['state']))
    errs = [float(np.abs(p[:, ::2, ::2].numpy() - g[f'prob_sub_{t}']).max()) for t, p in enumerate(outs)]
    print('detections clip: max-abs prob err per frame', ['%.1e' % e for e in errs])
    assert max(errs) <= 1e-3
    for t, p in enumerate(outs):
        # no floor run on this clip: argmax-identical at every pixel with a reference margin above the fixed 2 x 1e-3
        assert _margin_aware_mismatch(p[:, ::2, ::2], torch.from_numpy(g[f'prob_sub_{t}'])) == 0, t


@pytest.mark.parametrize('mode', ['queries', 'owner', 'bank'])
def test_sharded_read_single_rank_group_is_identical(network, mode):
    """The three one-clip-on-several-GPUs modes of MemoryManager over a 1-rank RCCL group: the collective
    code paths (all-gather / gather of read-out columns, all-reduce of usage; broadcast of query and
    memory rows in frame-owner mode; all-gather of candidate keys + all-reduce of partial read-outs in
    bank-sharded mode) must reproduce the plain read bit for bit (the 2- and 3-rank splits run on
    CPU/gloo in tests/test_sharded_read_gloo.py)"""
    import socket
    import torch.distributed as dist
    from deva.inference.inference_core import DEVAInferenceCore
    net = network
    sc = dict(scenarios.E2E['two_buckets'])
    sc['frames'] = 20
    plain, core_a = scenarios.run_scenario(lambda cfg: DEVAInferenceCore(net, cfg), sc, device=dev())
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        def make(cfg):
            c = DEVAInferenceCore(net, cfg)
            if mode == 'bank':
                c.memory.shard_bank()
            else:
                c.memory.shard_queries(owner=0 if mode == 'owner' else None)
            return c
        sharded, core_b = scenarios.run_scenario(make, sc, device=dev())
    finally:
        dist.destroy_process_group()
    assert all(torch.equal(a, b) for a, b in zip(plain, sharded))
    assert core_b.memory.comm_bytes >= 0  # bytes exchanged with OTHER ranks: none in a 1-rank group
    for b in core_a.memory.work_mem.buckets:
        assert torch.equal(core_a.memory.work_mem.get_usage(b), core_b.memory.work_mem.get_usage(b))


def test_spatial_alignment_against_reference_golden(network, golden_dir):
    """semi-online voting window: one-frame fused memory read + decoder"""
    from deva.inference.consensus_associated import spatial_alignment
    from deva.inference.image_feature_store import ImageFeatureStore
    g = torch.load(os.path.join(golden_dir, 'alignment.pt'))
    frames, masks = scenarios.alignment_inputs(scenarios.ALIGNMENT)
    store = ImageFeatureStore(network, no_warning=True)
    out = spatial_alignment(0, frames[0].to(dev()), masks[0].to(dev()), 1, frames[1].to(dev()), network, store,
                            synth.base_config())
    err = max_err(out, g['aligned'])
    print(f'spatial_alignment max abs err {err:.3e}')
    assert out.shape == g['aligned'].shape and err <= 1e-3


def test_consensus_auto_association_against_reference_golden(network, golden_dir):
    """semi-online voting with inferred association: joint-histogram IoU table (one launch per frame pair, one
    copy), matching, selection, painting vs the reference's find_consensus_auto_association"""
    from deva.inference.image_feature_store import ImageFeatureStore
    from deva.inference.object_info import ObjectInfo
    got = scenarios.run_consensus_cases(network, lambda: ImageFeatureStore(network, no_warning=True),
                                        lambda **kw: ObjectInfo(**kw), device=dev())
    scenarios.check_consensus_cases(got, torch.load(os.path.join(golden_dir, 'consensus_auto.pt')))


def test_edge_paths_against_reference_golden(network, golden_dir):
    """API edge paths (no memory yet, soft masks, cache control, empty detection round)"""
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    got = scenarios.run_edge_cases(lambda cfg: DEVAInferenceCore(network, cfg), device=dev(), make_info=ObjectInfo)
    g = torch.load(os.path.join(golden_dir, 'edge_cases.pt'))
    assert got.keys() == g.keys()
    for k in g:
        assert got[k].shape == g[k].shape, k
        assert max_err(got[k].float(), g[k].float()) <= 1e-3 * max(1.0, g[k].float().abs().max().item()), k


def test_chunked_objects_clip_matches_unchunked(network):
    """--chunk_size 2 with five objects (3 chunks, the last one ragged) through encode_mask / segment:
    the same clip as the unchunked run up to fp32 summation-order noise (the split-K factor of a
    layer depends on its batch)"""
    from deva.inference.inference_core import DEVAInferenceCore
    sc = dict(scenarios.E2E['five_obj'])
    plain, _ = scenarios.run_scenario(lambda cfg: DEVAInferenceCore(network, cfg), sc, device=dev())
    sc_chunked = dict(sc, cfg=dict(sc['cfg'], chunk_size=2))
    chunked, _ = scenarios.run_scenario(lambda cfg: DEVAInferenceCore(network, cfg), sc_chunked, device=dev())
    worst = max(max_err(a, b) for a, b in zip(chunked, plain))
    print(f'chunk_size=2 vs unchunked: max abs prob difference {worst:.3e}')
    assert worst <= 1e-3


def test_read_memory_against_reference_golden(network, golden_dir):
    """DEVA.read_memory (dense, full softmax) on the dense-similarity / column-softmax / GEMM kernels"""
    g = torch.load(os.path.join(golden_dir, 'read_memory.pt'))
    out = network.read_memory(**{k: v.to(dev()) for k, v in g['args'].items()})
    err = max_err(out, g['out'])
    print(f'read_memory max abs err {err:.3e}')
    assert out.shape == g['out'].shape and err <= 1e-4 * max(1.0, g['out'].abs().max().item())
