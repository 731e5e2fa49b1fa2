"""Lock-step teacher-forced comparison of every stage of every frame (used by the CPU host-logic
test with the emulated ops at a small size, and by the GPU test at full 480p size).

The CPU oracle drives a clip (memory frame every 2nd frame); each stage of the package under test --
key encoder, key projection, memory read against a bank fed with the oracle's keys/values, mask
decoder, value encoder -- is run on the oracle's OWN inputs for that frame.  With identical keys the
top-k sets must be identical, so the soft outputs have to agree to 1e-3 (observed ~1e-5)."""
import torch

from oracle import deva_oracle as O
from workload import synth


def run(network, P, H, W, no, frames, device, stage_tol=2e-4, logits_tol=1e-3, prob_tol=1e-3, stage_tols=None, **config):
    """network: one network, or {tag: network} -- several builds of the SAME weights (fp32, --f16_split, --f16_split
    --f16_split_key_encoder) held against ONE oracle pass: every build sees the oracle's inputs of every stage, so the CPU
    work (the expensive part at 1080p / 4K) is shared.  Returns the worst relative errors per stage ({tag: {...}} for a
    dict).  stage_tols: per-stage overrides of stage_tol ({'value': 3e-3, ...}), for modes whose stages differ in precision."""
    from deva.inference.memory_manager import MemoryManager
    nets = network if isinstance(network, dict) else {'': network}
    cfg = synth.base_config(**dict({'mem_every': 2}, **config))
    d = device
    stream = synth.FrameStream(H, W, seed=4)
    objs = list(range(1, no + 1))
    omem = O.OracleMemory(cfg)
    hmem = MemoryManager(cfg)  # (fed with the oracle's keys / values: the read does not depend on the network build)
    sens_o = torch.zeros(1, no, 512, H // 16, W // 16)
    prob_o = torch.softmax(O.aggregate(torch.stack([synth.box_mask(H, W, no) == o for o in objs], 0), 0), 0)
    worst_of = {tag: {} for tag in nets}

    def tracker(tag):
        worst = worst_of[tag]

        def track(name, got, ref, tol):
            e = (got.detach().cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            worst[name] = max(worst.get(name, 0.0), e)
            tol = (stage_tols or {}).get(name, tol)
            assert e <= tol, (tag, name, e, tol)
        return track

    for t in range(frames):
        img = stream.next().unsqueeze(0)
        ms_o, feat_o = O.encode_image(P, img)
        key_o, shr_o, sel_o = O.transform_key(P, feat_o)
        ms_od = tuple(x.to(d) for x in ms_o)
        for tag, net in nets.items():
            track = tracker(tag)
            ms_h, feat_h = net.encode_image(img.to(d))
            for n, a, b in zip(('f16', 'f8', 'f4'), ms_h, ms_o):
                track(n, a, b, stage_tol)
            track('feat', feat_h, feat_o, stage_tol)
            key_h, shr_h, sel_h = net.transform_key(feat_o.to(d))
            track('key', key_h, key_o, stage_tol)
            track('shrinkage', shr_h, shr_o, stage_tol)
            track('selection', sel_h, sel_o, stage_tol)
        if t > 0:
            ro_o = omem.match(key_o, sel_o)
            ro_h = hmem.match_memory(key_o.to(d), sel_o.to(d))
            ro_o = torch.stack([ro_o[o] for o in objs], 0).unsqueeze(0)
            ro_h = torch.stack([ro_h[o] for o in objs], 0).unsqueeze(0)
            for tag in nets:
                tracker(tag)('readout', ro_h, ro_o, 1e-4)
            last = prob_o[1:].unsqueeze(0)
            s_o, lg_o, pr_o = O.segment(P, ms_o, ro_o, sens_o, last)
            for tag, net in nets.items():
                track, worst = tracker(tag), worst_of[tag]
                s_h, lg_h, pr_h = net.segment(ms_od, ro_o.to(d), sens_o.to(d), last.to(d))
                track('sensory_seg', s_h, s_o, stage_tol)
                worst['logits_abs'] = max(worst.get('logits_abs', 0.0), (lg_h.cpu() - lg_o).abs().max().item())
                worst['prob_abs'] = max(worst.get('prob_abs', 0.0), (pr_h.cpu() - pr_o).abs().max().item())
                assert worst['logits_abs'] <= logits_tol, (tag, worst['logits_abs'])
                assert worst['prob_abs'] <= prob_tol, (tag, worst['prob_abs'])
                track('logits', lg_h, lg_o, stage_tol)
                track('prob', pr_h, pr_o, stage_tol)
            sens_o, prob_o = s_o, pr_o[0]
        if t % cfg['mem_every'] == 0:
            last = prob_o[1:].unsqueeze(0)
            v_o, s2_o = O.encode_mask(P, img, ms_o[0], sens_o, last)
            for tag, net in nets.items():
                track = tracker(tag)
                v_h, s2_h = net.encode_mask(img.to(d), ms_od, sens_o.to(d), last.to(d))
                track('value', v_h, v_o, stage_tol)
                track('sensory_deep', s2_h, s2_o, stage_tol)
            omem.add(key_o, shr_o, v_o, objs, sel_o)
            hmem.add_memory(key_o.to(d), shr_o.to(d), v_o.to(d), objs, selection=sel_o.to(d))
            sens_o = s2_o
    return worst_of if isinstance(network, dict) else worst_of['']
