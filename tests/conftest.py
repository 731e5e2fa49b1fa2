import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'tracking-anything-with-deva_amd')
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def recipe_state_dict():
    """Recipe weights (workload/weights.py) for the spec committed in tests/golden."""
    import json
    import torch
    from workload import weights
    with open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_spec.json')) as f:
        spec = json.load(f)
    triples = [(k, tuple(s), getattr(torch, d)) for k, s, d in spec['tensors']]
    return weights.make_state_dict(triples, seed=0), spec


@pytest.fixture(scope='session')
def peaky_state_dict(recipe_state_dict):
    """the second recipe (sharper affinities and logits; workload/weights.py:RECIPES['peaky'])"""
    import torch
    from workload import weights
    _, spec = recipe_state_dict
    triples = [(k, tuple(s), getattr(torch, d)) for k, s, d in spec['tensors']]
    return weights.make_state_dict(triples, seed=0, recipe='peaky')


def pytest_sessionstart(session):
    """DEVA_TEST_DRYRUN=1 (builder's container, no GPU): run the -m gpu test CODE on the CPU with the
    emulated ops so that a typo does not cost a GPU-box session.  Never set on the GPU box: there the
    -m gpu tests run the HIP library (the driver records which .so files the test process loaded)."""
    if os.environ.get('DEVA_TEST_DRYRUN') != '1':
        return
    import torch
    assert not torch.cuda.is_available(), 'DEVA_TEST_DRYRUN is for boxes without a GPU'
    import emu_ops
    import gpu_util

    class _Setter:
        @staticmethod
        def setattr(obj, name, value):
            setattr(obj, name, value)

    emu_ops.install(_Setter)
    gpu_util.dev = lambda: torch.device('cpu')
    torch.cuda.synchronize = lambda *a, **k: None
