"""The frame loops of the reference's evaluation drivers on libdeva_hip.so, on a box WITHOUT a reference checkout
(VERDICT r3 missing 1 / next 4): tests/driver_loops.py restates evaluation/eval_vos.py:133-184 and the semi-online
schedule of evaluation/eval_with_detections.py:150-297 against the public interface; here they run on the package --
`get_model_and_config(parser)` (argv like the drivers') -> `.cuda().eval()` network, DataLoader-shaped CPU tensors ->
`.cuda()` -> `processor.step` / `vote_in_temporary_buffer` / `incorporate_detection` -> argmax -> `tmp_to_obj_cls` ->
per-frame synchronize -> saver thread -- and the index masks they would write are compared with what the REFERENCE
wrote for the same inputs (tests/golden/e2e_vos_example.npz: example/vos; tests/golden/driver_semionline.npz:
tests/golden/make_golden.py:gen_driver_semionline).  tests/test_gpu_h_reference_drivers.py keeps the UNCHANGED
scripts for boxes that do have a checkout."""
import os
import sys
from argparse import ArgumentParser

import numpy as np
import pytest
import torch

import driver_loops
from gpu_util import dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
# flipped-pixel allowances measured for the unchanged drivers on the emulated ops (tests/test_reference_drivers_cpu.py)
MAX_FLIPPED_VOS, MAX_FLIPPED_DET = 0.002, 0.015


@pytest.fixture(scope='module')
def network(tmp_path_factory, recipe_state_dict):
    """the drivers' way to the network: parser -> get_model_and_config -> checkpoint file -> device"""
    from deva.inference.eval_args import add_common_eval_args, get_model_and_config
    sd, _ = recipe_state_dict
    ckpt = str(tmp_path_factory.mktemp('ckpt') / 'recipe.pth')
    torch.save(sd, ckpt)
    parser = ArgumentParser()
    add_common_eval_args(parser)
    parser.add_argument('--temporal_setting', default='semionline')   # eval_with_detections.py's own flags
    parser.add_argument('--num_voting_frames', type=int, default=3)
    parser.add_argument('--detection_every', type=int, default=5)
    parser.add_argument('--max_missed_detection_count', type=int, default=2)
    parser.add_argument('--max_num_objects', type=int, default=-1)
    argv = sys.argv
    sys.argv = ['eval_driver', '--model', ckpt, '--output', str(tmp_path_factory.mktemp('out')), '--size', '480', '--mem_every', '5']
    try:
        if os.environ.get('DEVA_TEST_DRYRUN') == '1':  # CPU dry run of this file: `.cuda()` is the one call that cannot run
            torch.nn.Module.cuda = lambda self, *a, **k: self
        net, config, args = get_model_and_config(parser)
    finally:
        sys.argv = argv
    return net, config, args


def test_eval_vos_loop_on_hip(network, golden_dir):
    from deva.inference.inference_core import DEVAInferenceCore
    net, config, _ = network
    g = np.load(os.path.join(golden_dir, 'e2e_vos_example.npz'))
    frames = torch.from_numpy(g['frames'])
    masks, seconds = driver_loops.eval_vos_loop(lambda c: DEVAInferenceCore(net, config=c), config, frames,
                                                torch.from_numpy(g['annotation']), device=dev())
    assert sorted(masks) == [f'{t:05d}' for t in range(frames.shape[0])]
    lut = np.array([0] + g['labels'].tolist())
    for t in range(frames.shape[0]):
        got = masks[f'{t:05d}'].numpy()
        assert got.shape == tuple(frames.shape[1:3])
        ref = g['prob_sub'][t]  # the reference's probabilities, every 4th pixel
        top2 = np.sort(ref, axis=0)[-2:]
        decisive = (top2[1] - top2[0]) > 1e-2
        want = lut[ref.argmax(0)]
        sub = got[::4, ::4]
        assert int(((sub != want) & decisive).sum()) == 0, f'frame {t}: a pixel decided by > 1e-2 differs'
        full = lut[g['argmax'][t]]
        flipped = float((got != full).mean())
        print(f'eval_vos loop frame {t}: {int((got != full).sum())} of {got.size} pixels differ from the reference argmax')
        assert flipped <= MAX_FLIPPED_VOS, (t, flipped)
    print(f'eval_vos loop: {frames.shape[0]} frames, {seconds * 1e3:.1f} ms inside the per-frame event pairs')


def test_eval_vos_loop_resizes_like_the_driver(network, golden_dir):
    """the need_resize branch (eval_vos.py:170-175): F.interpolate on the probabilities before the argmax"""
    from deva.inference.inference_core import DEVAInferenceCore
    net, config, _ = network
    g = np.load(os.path.join(golden_dir, 'e2e_vos_example.npz'))
    frames = torch.from_numpy(g['frames'])[:2]
    masks, _ = driver_loops.eval_vos_loop(lambda c: DEVAInferenceCore(net, config=c), config, frames,
                                          torch.from_numpy(g['annotation']), device=dev(), out_size=(240, 427))
    assert all(tuple(m.shape) == (240, 427) for m in masks.values())
    assert set(torch.unique(masks['00000']).tolist()) <= {0, *g['labels'].tolist()}


def test_semionline_loop_on_hip(network, golden_dir):
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    net, config, _ = network
    g = np.load(os.path.join(golden_dir, 'driver_semionline.npz'))
    frames, dets = driver_loops.semionline_clip()
    cfg = dict(config, mem_every=2, max_missed_detection_count=2, max_num_objects=-1, num_voting_frames=3)
    masks, alive = driver_loops.semionline_loop(lambda c: DEVAInferenceCore(net, config=c), cfg, frames, dets,
                                                lambda **kw: ObjectInfo(**kw), num_voting_frames=3, detection_every=5,
                                                device=dev())
    names = g['names'].tolist()
    assert sorted(masks) == names
    assert alive == g['alive'].tolist(), (alive, g['alive'].tolist())
    for i, n in enumerate(names):
        got, want = masks[n].numpy(), g['masks'][i]
        assert sorted(set(got.flatten().tolist())) == sorted(set(want.flatten().tolist())), f'frame {n}: object ids differ'
        flipped = float((got != want).mean())
        print(f'semi-online loop frame {n}: {int((got != want).sum())} of {got.size} pixels differ from the reference')
        assert flipped <= MAX_FLIPPED_DET, (n, flipped)
