"""The parity harness can FAIL (round 2's margin-aware argmax check could not): self-tests of
tests/memory_audit.py on hand-made cases -- a selection that differs at an exact tie is adopted, one that differs at
a real gap is reported and fails; a merged pixel at a near-tie is adopted, one at a decisive margin fails; `Drift`
rejects a soft error above 1e-3 and an argmax flip above the 2e-3 margin, and accepts what is inside."""
import pytest
import torch

import memory_audit
from oracle import deva_oracle as O
from workload import synth

torch.set_grad_enabled(False)
K = 30


def _read(tie: bool):
    """a bank / query set and the oracle's read of it; with tie=True rows (k-1) and k of query 0's ranking are made
    identical (an exact fp32 tie at the top-k boundary)"""
    mk, ms, qk, qe = synth.affinity_inputs(400, 24, seed=5)
    rows, shr = mk.t().contiguous(), ms.reshape(-1).clone()
    sim = O.get_similarity(rows.t().contiguous(), shr.view(1, -1), qk, qe)
    order = torch.topk(sim[:, 0], K + 1)[1]
    inside, outside = int(order[K - 1]), int(order[K])
    if tie:
        rows[outside], shr[outside] = rows[inside], shr[inside]
        sim = O.get_similarity(rows.t().contiguous(), shr.view(1, -1), qk, qe)
    idx = O.topk_softmax(sim, K)[0].t().contiguous()        # [hw, k] like the HIP read returns it
    return dict(mk=rows, ms=shr, qk=qk, qe=qe, idx=idx.clone(), k=K), sim, inside, outside


def test_tie_following_adopts_an_exact_tie_and_rejects_a_real_gap():
    hip, sim, inside, outside = _read(tie=True)
    ref_sel = hip['idx'][0].tolist()
    kept, other = (inside, outside) if inside in ref_sel else (outside, inside)
    hip['idx'][0, ref_sel.index(kept)] = other           # the "HIP run" breaks the tie the other way
    with memory_audit.TieFollowing('tie', [hip]) as tf:
        idx, w = O.topk_softmax(sim, K)
    tf.check()
    assert tf.adopted == 1 and other in idx[:, 0].tolist() and kept not in idx[:, 0].tolist()
    assert abs(float(w[:, 0].sum()) - 1.0) < 1e-6

    hip, sim, inside, outside = _read(tie=False)
    far = int(torch.topk(sim[:, 0], 200)[1][-1])           # a token 170 ranks below the boundary
    hip['idx'][0, 0] = far
    with memory_audit.TieFollowing('gap', [hip]) as tf:
        idx, _ = O.topk_softmax(sim, K)
    assert tf.adopted == 0 and far not in idx[:, 0].tolist()  # nothing adopted ...
    with pytest.raises(AssertionError, match='beyond a near-tie'):
        tf.check()                                            # ... and the difference is reported


def test_follow_merge_adopts_near_tie_pixels_only():
    prob = torch.tensor([[[0.5004, 0.9], [0.2, 0.1]], [[0.4996, 0.1], [0.8, 0.9]]])   # [2 labels, 2, 2]
    forward = prob.argmax(0)
    seen = {}
    real = O.merge_detection

    def probe(forward, *a, **kw):
        seen['forward'] = forward.clone()
        return None

    O.merge_detection = probe
    try:
        near = forward.clone()
        near[0, 0] = 1                                        # margin 8e-4 <= 2e-3: adopted
        with memory_audit.FollowMerge('near', lambda: prob, near) as fm:
            O.merge_detection(forward.clone(), None, None, None, None)
        fm.check()
        assert fm.adopted == 1 and torch.equal(seen['forward'], near)
        far = forward.clone()
        far[0, 1] = 1                                         # margin 0.8: a real disagreement
        with memory_audit.FollowMerge('far', lambda: prob, far) as fm:
            O.merge_detection(forward.clone(), None, None, None, None)
        assert torch.equal(seen['forward'], forward)
        with pytest.raises(AssertionError):
            fm.check()
    finally:
        O.merge_detection = real


def test_drift_fails_above_the_north_star_numbers():
    ref = torch.tensor([[[0.70, 0.5008]], [[0.30, 0.4992]]])   # reference margins 0.4 and 1.6e-3
    ok = memory_audit.Drift('ok')
    ok.add(ref + 5e-4, ref)
    tolerated = ref.clone()
    tolerated[:, 0, 1] = torch.tensor([0.4999, 0.5001])        # a flip inside the 1e-3 bound: only possible below the 2e-3 margin
    ok.add(tolerated, ref)
    report = ok.finish()
    assert report['raw_flips'] == 1 and report['decisive_flips'] == 0
    soft = memory_audit.Drift('soft')
    soft.add(ref + 2e-3, ref)
    with pytest.raises(AssertionError, match='exceeds'):
        soft.finish()
    # with a measured floor (peaky recipe) the soft bound may be several 1e-3 wide; the argmax margin stays 2e-3, so a
    # flip at a pixel the reference decides by 3e-3 fails even though the soft error is inside the bound
    wide = torch.tensor([[[0.5015, 0.7]], [[0.4985, 0.3]]])
    noisy = wide + torch.tensor([[[5e-4, 0.0]], [[-5e-4, 0.0]]])   # the reference's own drift: 5e-4 -> bound 5e-3
    got = wide.clone()
    got[:, 0, 0] = torch.tensor([0.4995, 0.5005])              # error 2e-3 <= 5e-3, but the argmax flips at margin 3e-3
    floor = memory_audit.Drift('floor', floor_bound=True)
    floor.add(got, wide, None, noisy)
    with pytest.raises(AssertionError, match='argmax flips'):
        floor.finish()
