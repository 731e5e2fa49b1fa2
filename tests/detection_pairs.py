"""Free-running HIP-vs-oracle comparison of clips that merge detections (BASELINE configs[2]), shared by the 96x128
clip that the reference itself recorded (tests/test_gpu_e_network.py) and the 1080p clips (tests/test_gpu_g_fullsize.py).

Per frame the HIP core runs first (its memory reads and, on detection frames, its forward pass are tapped); the CPU
oracle then runs the same frame under `TieFollowing` + `FollowMerge` (tests/memory_audit.py): identical arithmetic,
the HIP run's decisions adopted at measured fp32 near-ties only -- top-k selections whose score gap is within the
measured score noise, merged forward-argmax pixels whose reference margin is <= 2e-3.  Against that reference the
HIP outputs are held to the north-star numbers as written on every frame (1e-3 max-abs, argmax-identical above a
2e-3 margin); the merged hard masks and the object tables must be identical."""
import torch

import memory_audit
from gpu_util import dev
from oracle import deva_oracle as O
from workload import detections, synth


def run(tag, hip, orc, H, W, frames, every, detection_of, make_info, noisy=None, clean=None, prefill=None,
        same_ids=True, seed=7, golden=None):
    """detection_of(t) -> (mask, info) of detection frame t, or a `ConsistentDetector` whose detections are
    generated from the HIP run's forward masks (the HIP run defines the clip).  noisy / clean: optional plain oracles
    (1e-6-perturbed inputs / unperturbed) replaying the same detections, for the reported noise floor.
    golden(t) -> stored reference output of frame t (instead of `clean`)."""
    stream = synth.FrameStream(H, W, seed=seed)
    gen = torch.Generator().manual_seed(0)
    drift = memory_audit.Drift(tag)
    hip_seg, orc_seg = [], []
    hip_segment, orc_segment = hip._segment, orc._segment

    def tap(store, fn):
        def tapped(*a, **kw):
            store.append(fn(*a, **kw))
            return store[-1]
        return tapped

    hip._segment, orc._segment = tap(hip_seg, hip_segment), tap(orc_seg, orc_segment)
    generator = detection_of if isinstance(detection_of, detections.ConsistentDetector) else None
    recorded, adopted, merged_px = {}, 0, 0
    for t in range(frames):
        img = stream.next()
        img_n = img * (1 + 1e-6 * torch.randn(img.shape, generator=gen))
        is_det = t % every == 0
        hip_forward = None
        with memory_audit.ReadTap() as tap_reads:
            if not is_det:
                a = hip.step(img.to(dev()), end=(t == frames - 1)).cpu()
            elif generator is not None:
                with detections.record_on_package(hip, generator, make_info, recorded, lambda: t):
                    a = hip.incorporate_detection(img.to(dev()), torch.zeros(H, W, dtype=torch.long, device=dev()), []).cpu()
                m, info = recorded[t]
            else:
                m, info = detection_of(t)
                a = hip.incorporate_detection(img.to(dev()), m.to(dev()), [make_info(**i) for i in info]).cpu()
        if is_det and hip_seg:
            hip_forward = hip_seg[-1].argmax(0).cpu()  # padded, tmp ids: what the HIP merge used
        with memory_audit.TieFollowing(f'{tag} frame {t}', tap_reads.reads) as tf:
            if is_det:
                with memory_audit.FollowMerge(f'{tag} frame {t}', lambda: orc_seg[-1], hip_forward) as fm:
                    b = orc.incorporate_detection(img, m, info)
                fm.check()
                merged_px += fm.adopted
            else:
                b = orc.step(img, end=(t == frames - 1))
        tf.check()
        assert not tf.queue, (tag, t)
        adopted += tf.adopted
        c = d = None
        if clean is not None:
            c = clean.incorporate_detection(img, m, info) if is_det else clean.step(img, end=(t == frames - 1))
        if golden is not None:
            c = golden(t)
        if noisy is not None:
            d = noisy.incorporate_detection(img_n, m, info) if is_det else noisy.step(img_n, end=(t == frames - 1))
        if t == 0 and prefill is not None:
            prefill(hip, [core for core in (orc, clean, noisy) if core is not None])
        if is_det:
            # a detection frame returns the +-16 logits of the merged HARD masks (inference_core.py:192): identical
            # masks given the adopted near-tie pixels; its forward pass is compared like a propagated frame
            assert torch.equal(a.argmax(0), b.argmax(0)), f'{tag} frame {t}: merged masks differ from the tie-following reference'
            if hip_forward is not None:
                pad = O.pad_to_multiple(torch.zeros(1, H, W))[1]
                drift.add(O.unpad(hip_seg[-1].cpu(), pad), O.unpad(orc_seg[-1], pad), frame=t, adopted_so_far=adopted + merged_px)
            if c is not None:
                print(f'{tag} frame {t} (detection): merged masks differ from the clean reference at '
                      f'{int((a.argmax(0) != c.argmax(0)).sum())} pixels; forward-argmax pixels adopted so far {merged_px}')
        else:
            drift.add(a, b, c, d, frame=t, adopted_so_far=adopted + merged_px)
        hip_seg.clear()
        orc_seg.clear()
        assert hip.object_manager.num_obj == len(orc.table), t
    report = drift.finish()
    report['forward_argmax_pixels_adopted'] = merged_px
    om = hip.object_manager
    if same_ids:  # (colliding ids are re-drawn from np.random, object_manager.py:40-50: the runs share its state)
        assert [int(o.id) for o in om.obj_to_tmp_id] == [r['id'] for r in orc.table]
    assert [int(o.poke_count) for o in om.obj_to_tmp_id] == [r['poke'] for r in orc.table]
    assert [list(o.category_ids) for o in om.obj_to_tmp_id] == [r['cats'] for r in orc.table]
    mem = hip.memory
    assert {b_: mem.work_mem.size(b_) for b_ in mem.work_mem.buckets} == {b_: orc.memory.work.size(b_) for b_ in orc.memory.work.buckets}
    assert {b_: mem.long_mem.size(b_) for b_ in mem.long_mem.buckets} == {b_: orc.memory.long.size(b_) for b_ in orc.memory.long.buckets}
    return report, recorded


def prefill_10k(hip, oracles):
    """bucket 0 exists now: pre-fill the long-term bank through the stores' own add"""
    objs = [r['id'] for r in oracles[0].table]
    key, shr, vals = synth.prefill_bank(10000, objs, seed=1)
    hip.memory.long_mem.add(key.to(dev()), {o: v.to(dev()) for o, v in vals.items()}, shr.to(dev()),
                            selection=None, supposed_bucket_id=0)
    for core in oracles:
        core.memory.long.add(key, vals, shr, None, bucket_id=0)
