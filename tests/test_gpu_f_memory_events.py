"""Teacher-forced consolidation / eviction / usage parity on the MI355X.

The CPU oracle drives the `lt_evict` and `two_buckets` clips; every memory call it makes is mirrored
into a HIP `MemoryManager` fed with the ORACLE'S keys / values (tests/memory_audit.py).  After every
read and every add -- including the three consolidations of each clip and the least-usage eviction
of `lt_evict` -- the HIP banks must hold exactly the oracle's tokens in the oracle's order
(bit-equal key rows <=> identical prototype index lists and identical surviving token sets), with
prototype values / shrinkage and the use / life counters (2^-40 fixed point here, fp32 sums in the
reference) within 1e-5, identical top-k selections and read-outs within 1e-5.
Reference: memory_manager.py:209-276, kv_memory_store.py:164-185, memory_utils.py:67-74."""
import json

import pytest
import torch

import memory_audit
import scenarios
from gpu_util import dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize('name', ['lt_evict', 'two_buckets'])
def test_memory_events_teacher_forced(recipe_state_dict, name):
    P, _ = recipe_state_dict
    report = memory_audit.teacher_forced_memory(P, scenarios.E2E[name], dev())
    print(f'{name} teacher-forced memory path:', json.dumps({k: float(f'{v:.3e}') for k, v in report.items()}))
    assert report['consolidations'] == 3
    assert report['evictions'] == (1 if name == 'lt_evict' else 0)
    assert report['tie_swapped_queries'] == 0
