"""Parity at the sizes bench.py quotes: BASELINE configs[2] (1080p, detections,
10 000-token long-term bank), the affinity kernel at the bench shapes, and a 4K lockstep.

* `test_affinity_at_bench_shapes`: the fused kernel at (N, HW) = (10 000, 8 160), (83 440, 8 160),
  (50 000, 32 400) against `get_similarity` + `topk` of the oracle, query-chunked on the CPU
  (memory_utils.py:6-76): every query at the first two shapes (usage counters included), a
  2 048-query subsample at the third.  This gates bench.py's `affinity` object.
* `test_1080p_detections_10k_bank_against_oracle`: 1088x1920 frames, detections every 5th frame
  through `incorporate_detection`, long-term bank pre-filled to 10 000 tokens through the store's own
  `add`, 12 frames with memory adds on propagated frames, HIP vs the CPU oracle
  (inference_core.py:137-198, memory_manager.py:91-169).  Gates `extra.fps_1080p...`.
* `test_4k_lockstep`: two teacher-forced 2160x3840 frames (32 400 queries).  Gates `extra.fps_4k...`.
"""
import json
import os

import pytest
import torch

import memory_audit
from gpu_util import dev, to_dev
from deva.hip import ops
from oracle import deva_oracle as O
from workload import synth

pytestmark = pytest.mark.gpu
# the builder's CPU dry run of this file (DEVA_TEST_DRYRUN=1, emulated ops) shrinks the frames; never set on the GPU box
FULL_HD = (144, 256) if os.environ.get('DEVA_TEST_DRYRUN') == '1' else (1080, 1920)
torch.set_grad_enabled(False)


@pytest.fixture(scope='module')
def network(recipe_state_dict):
    from deva.model.network import DEVA
    sd, _ = recipe_state_dict
    net = DEVA(synth.base_config())
    net.load_weights(sd)
    return net.to(dev()).eval()


@pytest.fixture(scope='module')
def peaky_network(peaky_state_dict):
    from deva.model.network import DEVA
    net = DEVA(synth.base_config())
    net.load_weights(peaky_state_dict)
    return net.to(dev()).eval()


@pytest.mark.parametrize('n,hw,cols', [(10000, 8160, None), (83440, 8160, None), (50000, 32400, 2048)])
def test_affinity_at_bench_shapes(n, hw, cols):
    k = 30
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=n + hw, key_scale=1.0)  # SURVEY §8d microbench inputs
    rows = to_dev(mk.t().contiguous())
    shr = to_dev(ms.reshape(-1).contiguous())
    fix = torch.zeros(n, dtype=torch.int64, device=dev())
    idx, w = ops.affinity_topk(None, None, 0, rows, shr, n, to_dev(qk), to_dev(qe), k, fix)
    torch.cuda.synchronize()
    idx, w = idx.cpu().long(), w.cpu()
    usage = (fix.cpu().double() / 2**40)
    pick = torch.arange(hw) if cols is None else torch.linspace(0, hw - 1, cols).long()
    ties = 0
    werr = 0.0
    ref_usage = torch.zeros(n, dtype=torch.float64)
    for lo in range(0, pick.numel(), 1024):
        q = pick[lo:lo + 1024]
        sim = O.get_similarity(mk, ms, qk[:, q], qe[:, q])          # [n, chunk]
        vals, ridx = torch.topk(sim.t().contiguous(), k=k + 1, dim=1)  # [chunk, k+1]
        got = idx[q]
        same_order = (got == ridx[:, :k]).all(1)
        same_set = (torch.sort(got, 1)[0] == torch.sort(ridx[:, :k], 1)[0]).all(1)
        for j in torch.nonzero(~same_set).flatten().tolist():
            gap = (vals[j, k - 1] - vals[j, k]).abs().item()
            assert gap <= 1e-6 * vals[j, k - 1].abs().item(), f'query {int(q[j])}: set differs at a non-tie (gap {gap:.3e})'
            ties += 1
        rw = vals[:, :k].exp()
        rw = rw / rw.sum(1, keepdim=True)
        sane = same_order & (vals[:, 0] > -80.0)
        if sane.any():
            werr = max(werr, (w[q][sane] - rw[sane]).abs().max().item())
        ref_usage.index_add_(0, ridx[:, :k].reshape(-1), torch.nan_to_num(rw).reshape(-1).double())
    print(f'affinity N={n} HW={hw}: {pick.numel()} queries checked, tie-swapped {ties}, weight err {werr:.2e}')
    assert werr <= 1e-5
    if cols is None and ties == 0:
        uerr = (usage - ref_usage).abs().max().item()
        print(f'  usage counters (fixed point vs fp64 sum of the reference weights): max abs err {uerr:.2e}')
        assert uerr <= 1e-4


def _paired_detection_clip(tag, hip, orc, noisy, H, W, frames, every, detect, prefill=None, strict=False,
                           same_ids=True):
    """HIP core, CPU oracle and the oracle on 1e-6-perturbed frames (the noise floor) side by side.
    detect(t, orc) -> (mask, info, b): runs the ORACLE's incorporate_detection of frame t (it may generate the
    detection from its own forward mask, workload/detections.py) and returns what it merged and its output;
    the same detection is then fed to the HIP core and to the perturbed oracle through the public interface.
    Propagated frames: soft outputs under Drift's three-tier rule; detection frames: the forward pass
    (inference_core.py:164-166) is compared like a propagated frame, and the merged hard masks may differ
    only at pixels where the two forward argmax differ (which Drift then holds to the margin rule)."""
    from deva.inference.object_info import ObjectInfo
    stream = synth.FrameStream(H, W, seed=7)
    gen = torch.Generator().manual_seed(0)
    drift = memory_audit.Drift(tag, strict=strict)
    seg_out = []
    hip_segment = hip._segment

    def tapped_segment(*args, **kw):
        seg_out.append(hip_segment(*args, **kw))
        return seg_out[-1]

    hip._segment = tapped_segment
    orc_pad = O.pad_to_multiple(torch.zeros(1, H, W))[1]
    for t in range(frames):
        img = stream.next()
        img_n = img * (1 + 1e-6 * torch.randn(img.shape, generator=gen))
        is_det = t % every == 0
        with memory_audit.ReadTap() as hip_tap, memory_audit.OracleTap() as ref_tap:
            if is_det:
                m, info, b = detect(t, img)
                a = hip.incorporate_detection(img.to(dev()), m.to(dev()), [ObjectInfo(**i) for i in info])
            else:
                a, b = hip.step(img.to(dev())), orc.step(img)
        if noisy is not None:
            c = noisy.incorporate_detection(img_n, m, info) if is_det else noisy.step(img_n)
        if t == 0 and prefill is not None:
            prefill(hip, orc, noisy)
        drift.audit_reads(t, hip_tap.reads, ref_tap.reads)
        if is_det and t > 0:
            fwd_h, fwd_o = O.unpad(seg_out[-1].cpu(), orc_pad), orc.trace['forward_prob']
            drift.add(fwd_h, fwd_o, None if noisy is None else noisy.trace['forward_prob'], frame=t)
            differ = (a.cpu().argmax(0) != b.argmax(0))
            fwd_differ = (fwd_h.argmax(0) != fwd_o.argmax(0))
            print(f'frame {t} (detection): merged masks differ at {int(differ.sum())} pixels, forward argmax at '
                  f'{int(fwd_differ.sum())}')
            assert int((differ & ~fwd_differ).sum()) == 0, t
            if int(differ.sum()):
                drift.note_flip(t)
        elif t == 0:
            assert (a.cpu() - b).abs().max().item() <= 1e-3  # nothing propagated yet: the detection itself
        else:
            drift.add(a.cpu(), b, None if noisy is None else c, frame=t)
        assert hip.object_manager.num_obj == len(orc.table), t
        del hip_tap, ref_tap
    report = drift.finish()
    om = hip.object_manager
    if same_ids:  # (colliding ids are re-drawn from np.random, object_manager.py:40-50: the two runs share its state)
        assert [int(o.id) for o in om.obj_to_tmp_id] == [r['id'] for r in orc.table]
    assert [int(o.poke_count) for o in om.obj_to_tmp_id] == [r['poke'] for r in orc.table]
    assert [[c for c in o.category_ids] for o in om.obj_to_tmp_id] == [r['cats'] for r in orc.table]
    mem = hip.memory
    assert {b: mem.work_mem.size(b) for b in mem.work_mem.buckets} == {b: orc.memory.work.size(b) for b in orc.memory.work.buckets}
    assert {b: mem.long_mem.size(b) for b in mem.long_mem.buckets} == {b: orc.memory.long.size(b) for b in orc.memory.long.buckets}
    return drift, report


def _prefill_10k(hip, orc, noisy):
    """bucket 0 exists now: pre-fill the long-term bank through the stores' own add"""
    objs = [r['id'] for r in orc.table]
    key, shr, vals = synth.prefill_bank(10000, objs, seed=1)
    hip.memory.long_mem.add(key.to(dev()), {o: v.to(dev()) for o, v in vals.items()}, shr.to(dev()),
                            selection=None, supposed_bucket_id=0)
    for core in (orc, noisy):
        if core is not None:
            core.memory.long.add(key, vals, shr, None, bucket_id=0)


def test_1080p_detections_10k_bank_against_oracle(network, recipe_state_dict):
    """the north-star target line's clip: ONE object, a fixed-box detection every 5th frame (with recipe weights
    it never matches, and --max_num_objects 1 discards it: the merge is a no-op), 10 000-token bank; the
    reference's own drift under a 1e-6 input perturbation is measured beside the HIP error"""
    from deva.inference.inference_core import DEVAInferenceCore
    P, _ = recipe_state_dict
    (H, W), frames, every = FULL_HD, 12, 5
    cfg = synth.base_config(mem_every=3, max_missed_detection_count=5, max_num_objects=-1)
    hip, orc, noisy = DEVAInferenceCore(network, cfg), O.OracleDetectionCore(P, cfg), O.OracleDetectionCore(P, cfg)

    def detect(t, img):
        m, info = synth.detection_frame(H, W, t, segments=1)
        return m, info, orc.incorporate_detection(img, m, info)

    drift, report = _paired_detection_clip('1080p/detections/10k-bank', hip, orc, noisy, H, W, frames, every, detect,
                                           prefill=_prefill_10k, same_ids=False)
    assert hip.memory.long_mem.size(0) == orc.memory.long.size(0) == 10000
    print('1080p detections clip:', json.dumps({k: float(f'{v:.3g}') for k, v in report.items()}))


def test_1080p_eight_segment_detections_against_oracle(peaky_network, peaky_state_dict):
    """BASELINE configs[2] as SURVEY.md 8d defines it, at 1080p: tracker-consistent detections
    (workload/detections.py) every 3rd frame -- re-detections that match and merge, new segments that spawn
    objects in new buckets, unseen objects that are purged --, long-term bank pre-filled to 10 000 tokens,
    >= 3 live objects throughout.  3 segments per detection here (the CPU oracle costs ~2.5 s per object and
    1080p frame; bench.py times the 8-segment clip, the 96x128 golden of the reference covers 4 segments and
    17 frames).  Peaky recipe: the reference's noise floor is ~3e-4 there, so the criteria bite."""
    from deva.inference.inference_core import DEVAInferenceCore
    from workload import detections
    P = peaky_state_dict
    (H, W), frames, every = FULL_HD, 8, 3
    cfg = synth.base_config(mem_every=2, max_missed_detection_count=1, max_num_objects=-1)
    hip, orc, noisy = DEVAInferenceCore(peaky_network, cfg), O.OracleDetectionCore(P, cfg), O.OracleDetectionCore(P, cfg)
    detector = detections.ConsistentDetector(H, W, segments=3, new_per_frame=1)
    pad = O.pad_to_multiple(torch.zeros(1, H, W))[1]
    recorded = {}

    def detect(t, img):
        with detections.record_on_oracle(O, detector, recorded, lambda: t, lambda: pad):
            b = orc.incorporate_detection(img, torch.zeros(H, W, dtype=torch.long), [])
        return (*recorded[t], b)

    drift, report = _paired_detection_clip('1080p/consistent detections', hip, orc, noisy, H, W, frames, every, detect,
                                           prefill=_prefill_10k)
    live = [len(info) for _, info in recorded.values()]
    assert hip.object_manager.num_obj >= 3 and len(hip.memory.work_mem.buckets) >= 2, (live, hip.object_manager.num_obj)
    assert any(i['id'] > 100000 for _, info in recorded.values() for i in info), 'no re-detection was generated'
    print('1080p consistent-detection clip:', json.dumps({k: float(f'{v:.3g}') for k, v in report.items()}),
          'objects at the end', [int(o.id) for o in hip.object_manager.obj_to_tmp_id])


def test_4k_lockstep(network, recipe_state_dict):
    import lockstep
    P, _ = recipe_state_dict
    worst = lockstep.run(network, P, 2160, 3840, 1, 2, dev())
    print('lockstep 4K worst relative errors:', json.dumps({k: float(f'{v:.2e}') for k, v in worst.items()}))
