"""The network stages and the full frame loop on the MI355X against (a) the golden vectors produced
by the reference itself and (b) the CPU oracle run side by side on the GPU box.

Tolerances (BASELINE.json north_star): soft outputs within 1e-3 max-abs of the reference CPU path;
hard masks argmax-identical wherever the oracle's top-1/top-2 probability margin exceeds twice the
observed max-abs difference (margin-aware rule of SURVEY.md §7: with synthetic weights many pixels
sit at ties that flip under any reduction-order change, including the reference against itself).
Teacher-forced per-stage checks use a much tighter 2e-4 relative bound."""
import json
import os

import numpy as np
import pytest
import torch

import scenarios
from gpu_util import dev, max_err
from oracle import deva_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope='module')
def network(recipe_state_dict):
    from deva.model.network import DEVA
    sd, _ = recipe_state_dict
    net = DEVA(synth.base_config())
    net.load_weights(sd)
    return net.to(dev()).eval()


def _margin_aware_mismatch(got: torch.Tensor, ref: torch.Tensor, err: float) -> int:
    """argmax mismatches at pixels whose reference top-1/top-2 margin exceeds 2*err"""
    top2 = ref.topk(2, dim=0)[0]
    decisive = (top2[0] - top2[1]) > 2 * err
    return int(((got.argmax(0) != ref.argmax(0)) & decisive).sum())


def _free_running_check(tag, got: torch.Tensor, ref: torch.Tensor):
    """Free-running clips at full resolution: a near-tied top-k decision can legitimately flip when
    the keys differ in the last bits (SURVEY.md §7 -- the reference does the same against itself),
    which moves the read-out of that one query by percents and shows up as an isolated spike.  So:
    report the max, require that all but a vanishing fraction of the soft outputs are within 1e-3,
    and require argmax identity wherever the reference's margin is decisive."""
    d = (got - ref).abs()
    err = d.max().item()
    frac = (d > 1e-3).float().mean().item()
    bad = _margin_aware_mismatch(got, ref, err)
    flips = int((got.argmax(0) != ref.argmax(0)).sum())
    print(f'{tag}: max-abs {err:.2e}, p99.99 {d.flatten().kthvalue(int(d.numel() * 0.9999))[0].item():.2e}, '
          f'frac>1e-3 {frac:.2e}, raw argmax flips {flips}/{got.shape[1] * got.shape[2]}, margin-aware mismatches {bad}')
    assert frac <= 1e-3, (tag, frac)
    assert err <= 2e-2, (tag, err)
    assert bad == 0, (tag, bad)


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
def test_e2e_against_reference_golden(network, golden_dir, name):
    from deva.inference.inference_core import DEVAInferenceCore
    sc = scenarios.E2E[name]
    outs, core = scenarios.run_scenario(lambda cfg: DEVAInferenceCore(network, cfg), sc, device=dev())
    g = np.load(os.path.join(golden_dir, f'e2e_{name}.npz'))
    assert [p.shape[0] for p in outs] == g['nchan'].tolist()
    sizes = json.loads(str(g['sizes']))
    mem = core.memory
    assert {str(b): mem.work_mem.size(b) for b in mem.work_mem.buckets} == sizes['work']
    if mem.use_long_term:
        assert {str(b): mem.long_mem.size(b) for b in mem.long_mem.buckets} == sizes['long']
    errs = [float(np.abs(p[:, ::2, ::2].numpy() - g[f'prob_sub_{t}']).max()) for t, p in enumerate(outs)]
    flips = [int((p.argmax(0).numpy() != g['argmax'][t]).sum()) for t, p in enumerate(outs)]
    print(f'{name}: max-abs prob err per frame {["%.1e" % e for e in errs]}')
    print(f'{name}: raw argmax flips per frame {flips} of {outs[0].shape[1] * outs[0].shape[2]} px')
    assert max(errs) <= 1e-3, (name, max(errs))


def test_vos_example_against_reference_golden(network, golden_dir):
    """BASELINE config 1: example/vos bmx-trees, 854x480 real frames, 2 objects, default flags"""
    from deva.inference.inference_core import DEVAInferenceCore
    g = np.load(os.path.join(golden_dir, 'e2e_vos_example.npz'))
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    core = DEVAInferenceCore(network, synth.base_config(enable_long_term_count_usage=False))
    labels = g['labels'].tolist()
    n = g['frames'].shape[0]
    for t in range(n):
        img = ((torch.from_numpy(g['frames'][t]).permute(2, 0, 1).float() / 255 - mean) / std).to(dev())
        if t == 0:
            p = core.step(img, torch.from_numpy(g['annotation'].astype(np.int64)).to(dev()), labels)
        else:
            p = core.step(img, end=(t == n - 1))
        p = p.cpu()
        _free_running_check(f'vos example frame {t}', p[:, ::4, ::4], torch.from_numpy(g['prob_sub'][t]))


def test_480p_five_objects_against_oracle(network, recipe_state_dict):
    """BASELINE config 2 shape (480x854 -> 480x864, 5 objects, working memory only), 7 frames,
    HIP runtime vs the CPU oracle on identical inputs."""
    from deva.inference.inference_core import DEVAInferenceCore
    P, _ = recipe_state_dict
    cfg = synth.base_config(enable_long_term=False, enable_long_term_count_usage=False)
    H, W, no, frames = 480, 854, 5, 7
    hip, orc = DEVAInferenceCore(network, cfg), O.OracleCore(P, cfg)
    stream = synth.FrameStream(H, W, seed=2)
    mask0 = synth.box_mask(H, W, no)
    objs = list(range(1, no + 1))
    for t in range(frames):
        img = stream.next()
        if t == 0:
            a, b = hip.step(img.to(dev()), mask0.to(dev()), objs), orc.step(img, mask0, objs)
        else:
            a, b = hip.step(img.to(dev())), orc.step(img)
        _free_running_check(f'480p/5obj frame {t}', a.cpu(), b)


def test_480p_lockstep_teacher_forced(network, recipe_state_dict):
    """Every stage of every frame at full 480x864 size on IDENTICAL inputs (tests/lockstep.py)."""
    import lockstep
    P, _ = recipe_state_dict
    worst = lockstep.run(network, P, 480, 864, 2, 7, dev())
    print('lockstep 480p worst relative errors:', json.dumps({k: float(f'{v:.2e}') for k, v in worst.items()}))
