"""The reference's UNCHANGED drivers on libdeva_hip.so (VERDICT r2 missing 4): the two tests of
tests/test_reference_drivers_cpu.py, run with `-m gpu` on a box that has BOTH a GPU and a reference checkout
(`DEVA_REFERENCE_ROOT=/path/to/Tracking-Anything-with-DEVA`).  tests/run_reference_driver.py emulates the ops only
when no GPU is visible, so here `evaluation/eval_vos.py` / `evaluation/eval_with_detections.py` drive the HIP
kernels and the expected outputs are the same (the reference's stored probabilities of example/vos; the reference
alone on example/vipseg).  The GPU boxes of this project carry no reference checkout: there the tests are SKIPPED
and this leg stays unverified on hardware (INTEGRATION.md says so)."""
import os

import pytest
import torch

import test_reference_drivers_cpu as cpu_leg

REF = os.environ.get('DEVA_REFERENCE_ROOT', '/root/reference')
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'evaluation')),
                                 reason='needs a reference checkout next to the GPU (DEVA_REFERENCE_ROOT)'),
              pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a HIP device')]

checkpoint = cpu_leg.checkpoint


def test_eval_vos_unchanged_on_hip(tmp_path, checkpoint, golden_dir):
    cpu_leg.test_eval_vos_unchanged_on_the_vos_example(tmp_path, checkpoint, golden_dir)


def test_eval_with_detections_unchanged_on_hip(tmp_path, checkpoint):
    cpu_leg.test_eval_with_detections_unchanged_on_the_vipseg_example(tmp_path, checkpoint)
