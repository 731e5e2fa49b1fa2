"""Pins the CPU oracle (oracle/deva_oracle.py) to the golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import scenarios
from oracle import deva_oracle as O
from workload import synth

torch.set_grad_enabled(False)
TOL = 2e-5  # bit-identical in the build container; slack for other CPUs / thread counts


def test_recipe_weights_match_generator(recipe_state_dict):
    sd, spec = recipe_state_dict
    h = hashlib.sha256()
    for k, _, _ in spec['tensors']:
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    assert h.hexdigest() == spec['sha256_seed0']
    assert len(sd) == 420


def test_memory_ops(golden_dir):
    cases = torch.load(os.path.join(golden_dir, 'memory_ops.pt'))
    for name, c in cases.items():
        mk, ms, qk, qe = synth.affinity_inputs(c['n'], c['hw'], seed=c['seed'], key_scale=c['scale'])
        sim = O.get_similarity(mk, ms, qk, qe)
        assert torch.allclose(sim, c['sim'], rtol=1e-6, atol=1e-6), name
        idx, w = O.topk_softmax(sim, 30)
        assert torch.equal(idx.int(), c['topk_indices']), name
        aff, usage = O.dense_affinity(sim, 30)
        assert torch.allclose(usage, c['usage'], rtol=1e-6, atol=1e-7), name
        v = synth.value_inputs(2, 512, c['n'], seed=c['seed'])
        ro = (v.view(-1, c['n']) @ aff).view(2, 512, -1)
        assert torch.allclose(ro, c['readout'], rtol=1e-5, atol=1e-6), name
        full, _ = O.dense_affinity(sim, None)
        assert torch.allclose(full, c['full_softmax'], rtol=1e-6, atol=1e-9), name


def test_stages(golden_dir, recipe_state_dict):
    P, _ = recipe_state_dict
    g = torch.load(os.path.join(golden_dir, 'stages_96x128.pt'))
    H, W, no = 96, 128, 2
    img = synth.FrameStream(H, W, seed=5).next().unsqueeze(0)
    ms, feat = O.encode_image(P, img)
    key, shr, sel = O.transform_key(P, feat)
    masks, sensory, readout = synth.stage_inputs(H, W, no)
    value, sens_deep = O.encode_mask(P, img, ms[0], sensory, masks)
    sens_seg, logits, prob = O.segment(P, ms, readout, sensory, masks)
    got = dict(f16=ms[0], f8=ms[1], f4=ms[2], feat=feat, key=key, shrinkage=shr, selection=sel,
               value=value, sensory_deep=sens_deep, sensory_seg=sens_seg, logits=logits, prob=prob)
    for k, v in got.items():
        err = (v - g[k]).abs().max().item()
        assert err <= TOL * max(1.0, g[k].abs().max().item()), (k, err)


@pytest.mark.parametrize('name', list(scenarios.E2E))
def test_e2e(golden_dir, recipe_state_dict, name):
    P, _ = recipe_state_dict
    sc = scenarios.E2E[name]
    outs, core = scenarios.run_scenario(lambda cfg: O.OracleCore(P, cfg), sc)
    g = np.load(os.path.join(golden_dir, f'e2e_{name}.npz'))
    assert [p.shape[0] for p in outs] == g['nchan'].tolist()
    for t, p in enumerate(outs):
        err = np.abs(p[:, ::2, ::2].numpy() - g[f'prob_sub_{t}']).max()
        assert err <= 1e-4, (name, t, err)
    sizes = json.loads(str(g['sizes']))
    assert {str(b): core.memory.work.size(b) for b in core.memory.work.buckets} == sizes['work']
    if core.memory.long is not None:
        assert {str(b): core.memory.long.size(b) for b in core.memory.long.buckets} == sizes['long']


def test_vos_example(golden_dir, recipe_state_dict):
    """BASELINE config 1: real frames (example/vos bmx-trees), 2 objects, default flags."""
    P, _ = recipe_state_dict
    g = np.load(os.path.join(golden_dir, 'e2e_vos_example.npz'))
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    core = O.OracleCore(P, synth.base_config(enable_long_term_count_usage=False))
    labels = g['labels'].tolist()
    n = 2  # two frames are enough on the CPU (each costs ~1.5 s)
    for t in range(n):
        img = (torch.from_numpy(g['frames'][t]).permute(2, 0, 1).float() / 255 - mean) / std
        if t == 0:
            p = core.step(img, torch.from_numpy(g['annotation'].astype(np.int64)), labels)
        else:
            p = core.step(img)
        err = np.abs(p[:, ::4, ::4].numpy() - g['prob_sub'][t]).max()
        assert err <= 1e-4, (t, err)
        assert (p.argmax(0).numpy() != g['argmax'][t]).mean() < 1e-4


def test_detection_restatement_against_reference_golden(golden_dir, recipe_state_dict):
    """OracleDetectionCore (incorporate_detection + match/merge restated on a plain object table)
    against the reference's outputs and final ObjectManager state on the 13-frame detection clip"""
    import json
    P, _ = recipe_state_dict
    outs, core = scenarios.run_detection_scenario(lambda cfg: O.OracleDetectionCore(P, cfg),
                                                  lambda **kw: dict(kw), scenarios.DETECTION)
    g = np.load(os.path.join(golden_dir, 'e2e_detections.npz'))
    assert [p.shape[0] for p in outs] == g['nchan'].tolist()
    want = json.loads(str(g['state']))
    assert [r['id'] for r in core.table] == want['ids']
    assert [r['poke'] for r in core.table] == want['poke']
    assert [r['cats'] for r in core.table] == want['cats']
    assert [r['isthing'] for r in core.table] == want['isthing']
    for t, p in enumerate(outs):
        assert np.abs(p[:, ::2, ::2].numpy() - g[f'prob_sub_{t}']).max() <= 1e-5, t


def test_e2e_peaky_recipe(golden_dir, peaky_state_dict):
    """the oracle is pinned on the second weight recipe as well (the reference's own outputs)"""
    sc = scenarios.E2E_PEAKY['peaky']
    outs, core = scenarios.run_scenario(lambda cfg: O.OracleCore(peaky_state_dict, cfg), sc)
    g = np.load(os.path.join(golden_dir, 'e2e_peaky.npz'))
    assert [p.shape[0] for p in outs] == g['nchan'].tolist()
    for t, p in enumerate(outs):
        assert np.abs(p[:, ::2, ::2].numpy() - g[f'prob_sub_{t}']).max() <= 1e-4, t
        assert (p.argmax(0).numpy() != g['argmax'][t]).mean() < 1e-4


def test_consistent_detection_clip_against_reference_golden(golden_dir, recipe_state_dict):
    """tracker-consistent detections (workload/detections.py): the oracle GENERATES the clip from its own
    forward masks through the merge hook and must arrive at the detections, outputs, object table and bank
    sizes the reference arrived at (matches, new buckets, purges, consolidation of several buckets)"""
    from workload import detections
    sc = scenarios.CONSISTENT
    pad = O.pad_to_multiple(torch.zeros(1, sc['H'], sc['W']))[1]
    outs, core, recorded = scenarios.run_consistent_detection_scenario(
        lambda cfg: O.OracleDetectionCore(recipe_state_dict[0], cfg), lambda **kw: dict(kw), sc,
        record=lambda det, rec, frame_of: detections.record_on_oracle(O, det, rec, frame_of, lambda: pad))
    g, golden_dets = scenarios.load_consistent_golden(golden_dir)
    assert sorted(recorded) == sorted(golden_dets)
    for t, (m, info) in recorded.items():
        assert info == golden_dets[t][1], t
        assert torch.equal(m, golden_dets[t][0]), t
    assert [p.shape[0] for p in outs] == g['nchan'].tolist()
    want = json.loads(str(g['state']))
    assert [r['id'] for r in core.table] == want['ids']
    assert [r['poke'] for r in core.table] == want['poke']
    assert [r['cats'] for r in core.table] == want['cats']
    assert [r['isthing'] for r in core.table] == want['isthing']
    sizes = json.loads(str(g['sizes']))
    assert {str(b): core.memory.work.size(b) for b in core.memory.work.buckets} == sizes['work']
    assert {str(b): core.memory.long.size(b) for b in core.memory.long.buckets} == sizes['long']
    for t, p in enumerate(outs):
        assert np.abs(p[:, ::2, ::2].numpy() - g[f'prob_sub_{t}']).max() <= 1e-4, t
