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
    """Recipe weights (oracle/weights.py) for the spec committed in tests/golden."""
    import json
    import torch
    from oracle import weights
    with open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_spec.json')) as f:
        spec = json.load(f)
    triples = [(k, tuple(s), getattr(torch, d)) for k, s, d in spec['tensors']]
    return weights.make_state_dict(triples, seed=0), spec
