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

import pytest
import torch

import memory_audit
from gpu_util import dev, to_dev
from deva.hip import ops
from oracle import deva_oracle as O
from workload import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope='module')
def network(recipe_state_dict):
    from deva.model.network import DEVA
    sd, _ = recipe_state_dict
    net = DEVA(synth.base_config())
    net.load_weights(sd)
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


def test_1080p_detections_10k_bank_against_oracle(network, recipe_state_dict):
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    P, _ = recipe_state_dict
    H, W, frames, every = 1080, 1920, 12, 5
    cfg = synth.base_config(mem_every=3, max_missed_detection_count=5, max_num_objects=-1)
    hip, orc = DEVAInferenceCore(network, cfg), O.OracleDetectionCore(P, cfg)
    stream = synth.FrameStream(H, W, seed=7)
    drift = memory_audit.Drift('1080p/detections/10k-bank')
    seg_out = []  # forward passes of the HIP run
    hip_segment = hip._segment

    def tapped_segment(*args, **kw):
        seg_out.append(hip_segment(*args, **kw))
        return seg_out[-1]

    hip._segment = tapped_segment
    orc_pad = O.pad_to_multiple(torch.zeros(1, H, W))[1]
    for t in range(frames):
        img = stream.next()
        with memory_audit.ReadTap() as hip_tap, memory_audit.OracleTap() as ref_tap:
            if t % every == 0:
                m, info = synth.detection_frame(H, W, t, segments=1)
                a = hip.incorporate_detection(img.to(dev()), m.to(dev()), [ObjectInfo(**i) for i in info])
                b = orc.incorporate_detection(img, m, info)
            else:
                a, b = hip.step(img.to(dev())), orc.step(img)
        if t == 0:  # bucket 0 exists now: pre-fill the long-term bank through the stores' own add
            key, shr, vals = synth.prefill_bank(10000, [10], seed=1)
            hip.memory.long_mem.add(key.to(dev()), {o: v.to(dev()) for o, v in vals.items()}, shr.to(dev()),
                                    selection=None, supposed_bucket_id=0)
            orc.memory.long.add(key, vals, shr, None, bucket_id=0)
        drift.audit_reads(t, hip_tap.reads, ref_tap.reads)
        if t % every == 0 and t > 0:
            # a detection frame returns the logits of the merged HARD masks (inference_core.py:192); its
            # propagation half (inference_core.py:164-166) is compared like any propagated frame, and the
            # merged masks may differ only where the two forward passes' argmax differ (near-ties, which
            # the margin-aware rule inside drift.add has just checked)
            fwd_h, fwd_o = O.unpad(seg_out[-1].cpu(), orc_pad), orc.trace['forward_prob']
            drift.add(fwd_h, fwd_o, frame=t)
            differ = (a.cpu().argmax(0) != b.argmax(0))
            fwd_differ = (fwd_h.argmax(0) != fwd_o.argmax(0))
            print(f'frame {t} (detection): merged masks differ at {int(differ.sum())} pixels, forward argmax at '
                  f'{int(fwd_differ.sum())}')
            assert int((differ & ~fwd_differ).sum()) == 0, t
            if int(fwd_differ.sum()):
                drift.first_flip_frame = t if drift.first_flip_frame is None else drift.first_flip_frame
        elif t == 0:
            assert (a.cpu() - b).abs().max().item() <= 1e-3  # nothing propagated yet: the detection itself
        else:
            drift.add(a.cpu(), b, frame=t)
        del hip_tap, ref_tap
    drift.finish()
    mem = hip.memory
    assert mem.long_mem.size(0) == orc.memory.long.size(0) == 10000
    assert mem.work_mem.size(0) == orc.memory.work.size(0)
    worst = max(e for e, _ in drift.ours)
    assert drift.flips > 0 or worst <= 1e-3
    print(f'1080p detections clip: worst max-abs on propagated frames {worst:.2e}, hard decisions flipped at ties: {drift.flips}')


def test_4k_lockstep(network, recipe_state_dict):
    import lockstep
    P, _ = recipe_state_dict
    worst = lockstep.run(network, P, 2160, 3840, 1, 2, dev())
    print('lockstep 4K worst relative errors:', json.dumps({k: float(f'{v:.2e}') for k, v in worst.items()}))
