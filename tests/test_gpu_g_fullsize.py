"""Parity at the sizes bench.py quotes: BASELINE configs[2] (1080p, detections,
10 000-token long-term bank), the affinity kernel at the bench shapes, and a 4K lockstep.

* `test_affinity_at_bench_shapes`: the fused kernel at (N, HW) = (10 000, 8 160), (83 440, 8 160),
  (50 000, 32 400) against `get_similarity` + `topk` of the oracle, query-chunked on the CPU
  (memory_utils.py:6-76): every query at the first shape (usage counters included), 2 048-query
  subsamples at the other two (the CPU reference of all 83 440 x 8 160 scores costs a minute of the suite).  This gates bench.py's `affinity` object.
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
from gpu_util import dev, net_config, to_dev
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
    net = DEVA(net_config())
    net.load_weights(sd)
    return net.to(dev()).eval()


@pytest.fixture(scope='module')
def split_all_network(recipe_state_dict):
    """--f16_split --f16_split_key_encoder (the mode of bench.py's `fps_*_f16_split_key_encoder` lines)"""
    from deva.model.network import DEVA
    net = DEVA(net_config(f16_split=True, f16_split_key_encoder=True))
    net.load_weights(recipe_state_dict[0])
    return net.to(dev()).eval()


FULL_REPORT = os.environ.get('DEVA_TEST_FULL_REPORT') == '1'  # also run the plain / 1e-6-perturbed oracles (noise-floor report)


# every query at all three shapes (round 3 sampled 2 048 queries of the two larger ones; the GPU boxes' hosts compute the
# chunked CPU reference of all 83 440 x 8 160 scores in ~10 s)
@pytest.mark.parametrize('n,hw,cols', [(10000, 8160, None), (83440, 8160, None), (50000, 32400, None)])
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


def test_1080p_detections_10k_bank_against_oracle(network, recipe_state_dict):
    """the north-star target line's clip: ONE object, a fixed-box detection every 5th frame (with recipe weights it
    never matches: a second object spawns at each detection and is purged again), 10 000-token bank.  HIP vs the
    tie-following oracle under the north-star bound as written; beside it the clean oracle and the reference's own
    drift under a 1e-6 input perturbation (the noise floor the judge asked for at this size)"""
    import detection_pairs
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    P, _ = recipe_state_dict
    # 7 frames: detections at t = 0 and 5 (the second one merges into a propagated state), memory frames on propagated
    # frames at t = 3, 6 (round 5 ran 12 frames and a second, 1e-6-perturbed oracle: 280 s of the driver's 1 200 s limit;
    # DEVA_TEST_FULL_REPORT=1 still does, and profiles/r05/tests/test_gpu_g_fullsize.log holds that report -- the
    # reference's own drift on this clip 9.3e-3 beside HIP vs the tie-following oracle 2.0e-5)
    (H, W), frames, every = FULL_HD, (12 if FULL_REPORT else 7), 5
    cfg = synth.base_config(mem_every=3, max_missed_detection_count=5, max_num_objects=-1)
    hip = DEVAInferenceCore(network, cfg)
    orc = O.OracleDetectionCore(P, cfg)
    noisy = O.OracleDetectionCore(P, cfg) if FULL_REPORT else None
    report, _ = detection_pairs.run('1080p/detections/10k-bank', hip, orc, H, W, frames, every,
                                    lambda t: synth.detection_frame(H, W, t, segments=1), ObjectInfo, noisy=noisy,
                                    prefill=detection_pairs.prefill_10k, same_ids=False)
    assert hip.memory.long_mem.size(0) == orc.memory.long.size(0) == 10000
    print('1080p detections clip:', json.dumps({k: float(f'{v:.3g}') for k, v in report.items()}))


@pytest.mark.parametrize('build', ['fp32', 'f16_split+key_encoder'])
def test_1080p_eight_segment_detections_against_oracle(network, split_all_network, recipe_state_dict, build):
    """(fp32: the parity target; f16_split+key_encoder: the mode of bench.py's fastest 8-segment line, same bounds.)
    BASELINE configs[2] as SURVEY.md 8d defines it, at 1080p AND at the object count bench.py quotes its FPS at:
    tracker-consistent detections (workload/detections.py) with 8 segments every 2nd frame -- 8 new objects from the
    first detection, re-detections that match and merge plus 2 new objects (a new memory bucket) from the second --,
    long-term bank pre-filled to 10 000 tokens, >= 10 live objects on the last frames.  4 frames: the CPU oracle
    costs ~2.5 s per object and 1080p frame (bench.py times 25 frames of the same generator; the 96x128 golden of
    the reference covers 4 segments and 17 frames).  The HIP run defines the clip (its forward masks feed the
    detector).  Reference: inference_core.py:137-198, segment_merging.py:89-143."""
    import detection_pairs
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    from workload import detections
    P, _ = recipe_state_dict
    (H, W), frames, every = FULL_HD, 4, 2
    cfg = synth.base_config(mem_every=2, max_missed_detection_count=1, max_num_objects=-1)
    net = network if build == 'fp32' else split_all_network
    before = ops.split_fallbacks(dev())
    hip, orc = DEVAInferenceCore(net, cfg), O.OracleDetectionCore(P, cfg)
    detector = detections.ConsistentDetector(H, W, segments=8, new_per_frame=2)
    report, recorded = detection_pairs.run(f'1080p/8-segment detections [{build}]', hip, orc, H, W, frames, every, detector,
                                           ObjectInfo, prefill=detection_pairs.prefill_10k)
    assert ops.split_fallbacks(dev()) == before
    assert all(len(info) == 8 for _, info in recorded.values()), [len(info) for _, info in recorded.values()]
    assert hip.object_manager.num_obj >= 10 and len(hip.memory.work_mem.buckets) >= 2, hip.object_manager.num_obj
    assert any(i['id'] > 100000 for _, info in recorded.values() for i in info), 'no re-detection was generated'
    print(f'1080p 8-segment detection clip [{build}]:', json.dumps({k: float(f'{v:.3g}') for k, v in report.items()}),
          'objects at the end', [int(o.id) for o in hip.object_manager.obj_to_tmp_id])


def test_4k_free_running_50k_bank_against_oracle(network, recipe_state_dict):
    """BASELINE configs[4] on one GPU, FREE-RUNNING (round 3 had two teacher-forced frames only): 2160x3840, one
    object, long-term bank pre-filled to 50 000 tokens after the annotated frame, three frames -- the reads see
    50 000 + 32 400 (+ 32 400) tokens x 32 400 queries.  HIP vs the tie-following oracle under the north-star bound as
    written; the plain (clean) oracle and the reference's own drift under a 1e-6 input perturbation are printed beside
    it.  Reference: inference_core.py:200-290, memory_manager.py:91-169."""
    from deva.inference.inference_core import DEVAInferenceCore
    P, _ = recipe_state_dict
    dry = os.environ.get('DEVA_TEST_DRYRUN') == '1'
    (H, W), bank, frames = ((144, 256), 600, 3) if dry else ((2160, 3840), 50000, 3)
    cfg = synth.base_config(max_long_term_elements=bank, mem_every=2)
    hip = DEVAInferenceCore(network, cfg)
    following = O.OracleCore(P, cfg)
    # the clean oracle and the 1e-6-perturbed one cost two more minutes each at 4K: DEVA_TEST_FULL_REPORT=1 runs them
    # (profiles/r04/test_gpu_g_4k_free_running_full_report.log: vs clean 1.01e-2, the reference's own drift 1.11e-2)
    full = dry or os.environ.get('DEVA_TEST_FULL_REPORT') == '1'
    clean, noisy = (O.OracleCore(P, cfg), O.OracleCore(P, cfg)) if full else (None, None)
    gen = torch.Generator().manual_seed(0)
    stream = synth.FrameStream(H, W, seed=11)
    imgs = [stream.next() for _ in range(frames)]
    mask0 = synth.box_mask(H, W, 1)
    key, shr, vals = synth.prefill_bank(bank - cfg['num_prototypes'], [1], seed=2)

    def hip_step(t):
        out = hip.step(imgs[t].to(dev()), mask0.to(dev()) if t == 0 else None, [1] if t == 0 else None).cpu()
        if t == 0:  # bucket 0 exists now (pre-fill through the store's own add)
            hip.memory.long_mem.add(key.to(dev()), {o: v.to(dev()) for o, v in vals.items()}, shr.to(dev()),
                                    selection=None, supposed_bucket_id=0)
        return out

    def orc_step(core, perturb=False):
        def step(t):
            img = imgs[t] * (1 + 1e-6 * torch.randn(imgs[t].shape, generator=gen)) if perturb else imgs[t]
            out = core.step(img, mask0 if t == 0 else None, [1] if t == 0 else None)
            if t == 0:
                core.memory.long.add(key, vals, shr, None, bucket_id=0)
            return out
        return step

    report = memory_audit.paired_steps('4K/free-running/50k-bank', frames, hip_step, orc_step(following),
                                       orc_step(clean) if full else None, orc_step(noisy, perturb=True) if full else None)
    assert hip.memory.long_mem.size(0) == following.memory.long.size(0) == bank - cfg['num_prototypes']
    print('4K free-running clip:', json.dumps({k: float(f'{v:.3g}') for k, v in report.items()}))


def test_4k_lockstep(network, split_all_network, recipe_state_dict):
    """two teacher-forced 2160x3840 frames; fp32 and --f16_split --f16_split_key_encoder against ONE oracle pass (the CPU
    oracle is what a 4K frame costs), same bounds"""
    import lockstep
    P, _ = recipe_state_dict
    (H, W) = (144, 256) if os.environ.get('DEVA_TEST_DRYRUN') == '1' else (2160, 3840)
    before = ops.split_fallbacks(dev())
    worst = lockstep.run({'fp32': network, 'f16_split+key_encoder': split_all_network}, P, H, W, 1, 2, dev())
    for build, w in worst.items():
        print(f'lockstep 4K [{build}] worst relative errors:', json.dumps({k: float(f'{v:.2e}') for k, v in w.items()}))
    assert ops.split_fallbacks(dev()) == before
