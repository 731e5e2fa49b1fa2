"""Audit helpers for the memory path (shared by the CPU host-logic tests, which run them on the
emulated ops, and by the -m gpu tests, which run them on the HIP kernels).

1. `ReadTap` / `OracleTap`: record every top-k memory read of a HIP `MemoryManager` and of the CPU
   oracle (bank, queries, selected token lists).
2. `explain_flips`: compare the two runs' selections query by query.  A differing selection is
   *explained* only if the reference's own score gap between the tokens that were swapped is within
   the measured score noise between the two runs (their banks/queries differ in the last bits) --
   i.e. a near-tie that any last-bit change of the keys flips.  Everything else is a kernel bug.
3. `teacher_forced_memory`: drives a scenario with the CPU oracle and mirrors every memory call
   (match / add with its consolidation and eviction) into a HIP `MemoryManager` fed with the
   ORACLE'S inputs, asserting after every call: identical banks (keys bit-exact, i.e. identical
   prototype index lists and identical surviving token sets), prototype values / shrinkage and usage
   counters within 1e-5, identical top-k sets, read-outs within 1e-5 relative.
   Reference: memory_manager.py:91-276, kv_memory_store.py:35-185, memory_utils.py:48-76.
"""
from typing import Dict, List

import torch

from deva.hip import ops
from oracle import deva_oracle as O
from workload import synth


def _sim(mk_rows: torch.Tensor, ms: torch.Tensor, qk: torch.Tensor, qe: torch.Tensor) -> torch.Tensor:
    """reference similarity (memory_utils.py:6-45) of a token-major bank [N,64] on the CPU"""
    return O.get_similarity(mk_rows.t().contiguous(), ms.reshape(1, -1), qk, qe)


class ReadTap:
    """records the inputs and the selection of every `ops.affinity_topk` call"""

    def __init__(self):
        self.reads: List[Dict] = []
        self._orig = None

    def __enter__(self):
        self._orig = ops.affinity_topk
        tap = self

        def wrapped(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix=None, splits=None):
            idx, w = tap._orig(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix, splits)
            rows = [key_long[:n_long]] if n_long else []
            shr = [shr_long[:n_long]] if n_long else []
            rows.append(key_work[:n_work])
            shr.append(shr_work[:n_work])
            tap.reads.append(dict(mk=torch.cat(rows, 0).cpu(), ms=torch.cat(shr, 0).cpu(), qk=qk.cpu(), qe=qe.cpu(),
                                  idx=idx.cpu().long(), w=w.cpu(), n_long=n_long, k=k))
            return idx, w

        ops.affinity_topk = wrapped
        return self

    def __exit__(self, *exc):
        ops.affinity_topk = self._orig


class OracleTap:
    """records the similarity matrix and inputs of every top-k read of the oracle"""

    def __init__(self):
        self.reads: List[Dict] = []
        self._orig = None
        self._last_inputs = None

    def __enter__(self):
        self._orig = (O.get_similarity, O.dense_affinity)
        tap = self

        def get_similarity(mk, ms, qk, qe):
            tap._last_inputs = (mk, ms, qk, qe)
            return tap._orig[0](mk, ms, qk, qe)

        def dense_affinity(sim, k):
            if k is not None:
                mk, ms, qk, qe = tap._last_inputs
                tap.reads.append(dict(mk=mk.t().contiguous(), ms=ms.reshape(-1).clone(), qk=qk.clone(), qe=qe.clone(),
                                      sim=sim.clone(), k=k))
            return tap._orig[1](sim, k)

        O.get_similarity, O.dense_affinity = get_similarity, dense_affinity
        return self

    def __exit__(self, *exc):
        O.get_similarity, O.dense_affinity = self._orig


def explain_flips(tag: str, hip_read: Dict, ref_read: Dict, slack: float = 4.0):
    """-> (number of queries whose selected token set differs, the largest unexplained excess).
    A flip is explained when, for every token selected by one side only, the reference's distance of
    that token's score to its k-th/(k+1)-th boundary is <= slack x the measured score noise of the
    query (max |score from the HIP run's bank/queries - score from the reference's| over the tokens
    involved, both evaluated with the reference formula on the CPU)."""
    sim_ref = ref_read['sim']
    k = ref_read['k']
    n, hw = sim_ref.shape
    assert hip_read['mk'].shape[0] == n, f'{tag}: bank sizes differ ({hip_read["mk"].shape[0]} vs {n})'
    vals, ridx = torch.topk(sim_ref, k=min(k + 1, n), dim=0)
    ref_sets = ridx[:k].t()
    same = (torch.sort(hip_read['idx'], 1)[0] == torch.sort(ref_sets, 1)[0]).all(1)
    flipped = torch.nonzero(~same).flatten().tolist()
    worst_excess, lines = 0.0, []
    if flipped:
        sim_hip = _sim(hip_read['mk'], hip_read['ms'], hip_read['qk'], hip_read['qe'])
    for q in flipped:
        a, b = set(hip_read['idx'][q].tolist()), set(ref_sets[q].tolist())
        swapped = sorted(a ^ b)
        boundary = 0.5 * (vals[k - 1, q] + vals[k, q]).item() if n > k else vals[k - 1, q].item()
        gap = max(abs(sim_ref[t, q].item() - boundary) for t in swapped)
        noise = max(abs(sim_hip[t, q].item() - sim_ref[t, q].item()) for t in swapped)
        noise = max(noise, 1e-6 * abs(boundary))  # fp32 evaluation noise of the score itself
        excess = gap / noise
        worst_excess = max(worst_excess, excess)
        lines.append(f'{tag}: query {q}: {len(swapped) // 2} token(s) swapped, boundary score {boundary:.6g}, '
                     f'reference gap {gap:.3e}, measured score noise {noise:.3e} (ratio {excess:.2f})')
    for line in lines[:8]:
        print(line)
    return len(flipped), (worst_excess if flipped else 0.0), slack


NORTH_STAR = 1e-3  # BASELINE.json north_star: max-abs bound on the soft outputs


def margin_aware_mismatch(got: torch.Tensor, ref: torch.Tensor, floor: float = None) -> int:
    """argmax mismatches at pixels whose REFERENCE top-1/top-2 margin exceeds 2 x max(floor, 1e-3), where
    `floor` is the reference's own noise on this frame (reference vs the reference under a 1e-6 relative
    input perturbation) -- NOT the error under test (with the observed error as the margin
    the check cannot fail).  Without a floor run the margin is the fixed 2 x 1e-3 of the north-star bound."""
    return argmax_flips(got, ref, floor)[1]


def argmax_flips(got: torch.Tensor, ref: torch.Tensor, floor: float = None, margin: float = None):
    """-> (raw argmax flips, flips at pixels whose reference margin exceeds `margin`)
    margin defaults to 2 x max(floor, 1e-3)"""
    top2 = ref.topk(2, dim=0)[0]
    if margin is None:
        margin = 2 * max(floor or 0.0, NORTH_STAR)
    flipped = got.argmax(0) != ref.argmax(0)
    return int(flipped.sum()), int((flipped & ((top2[0] - top2[1]) > margin)).sum())


class Drift:
    """Free-running full-resolution clips, three tiers per frame (e = max-abs HIP vs reference):
      e <= 1e-3                      the north-star bound, always fine;
      1e-3 < e <= 10 x floor         allowed ONLY from a frame on in which a discrete decision of the HIP run
                                     differs from the reference's and is explained: a top-k selection at a
                                     measured near-tie (explain_flips: the reference's score gap at the
                                     k-th/(k+1)-th boundary is within the measured score noise) or a merged
                                     hard-mask pixel whose forward argmax differs (`note_flip`);
      e > max(1e-3, 10 x floor)      fails unconditionally.
    floor = the reference's own drift on this clip under a 1e-6 relative input perturbation (max over the
    frames), measured in the same test; without a floor run (or with strict=True) the bound is 1e-3 flat.
    Argmax: flips at pixels whose reference margin exceeds 2 x the bound must be zero (the bound does not
    depend on the error under test); raw flips, flips above 2 x max(frame floor, 1e-3) and the reference's
    own flips under the perturbation are printed per frame.  An unexplained differing top-k selection fails
    at once."""

    def __init__(self, tag, stride=1, strict=False):
        self.tag, self.ours, self.floor, self.stride, self.strict = tag, [], [], stride, strict
        self.frames = []
        self.pending = []  # (frame, got, ref) kept until the clip's floor is known
        self.first_flip_frame = None
        self.flips = 0
        self.raw_flips = self.flips_above_floor = self.ref_flips = 0

    @staticmethod
    def _stats(a, b):
        d = (a - b).abs()
        return d.max().item(), (d > 1e-3).float().mean().item()

    def note_flip(self, frame):
        """a discrete decision differed at `frame` (and was explained by the caller)"""
        if self.first_flip_frame is None:
            self.first_flip_frame = frame

    def audit_reads(self, frame, hip_reads, ref_reads):
        """compare the top-k selections of this frame's memory reads (one per bucket)"""
        assert len(hip_reads) == len(ref_reads), (self.tag, frame)
        for bi, (hr, rr) in enumerate(zip(hip_reads, ref_reads)):
            n, excess, slack = explain_flips(f'{self.tag} frame {frame} bucket#{bi}', hr, rr)
            assert excess <= slack, (f'{self.tag} frame {frame}: top-k selection differs from the reference and the '
                                     f'score gap is {excess:.1f}x the measured score noise: not a near-tie')
            if n:
                self.note_flip(frame)
            self.flips += n

    def add(self, got, ref, ref_perturbed=None, frame=None):
        frame = len(self.ours) if frame is None else frame
        err, frac = self._stats(got, ref)
        f_err = None
        msg = f'{self.tag} frame {frame}: HIP vs ref max-abs {err:.2e} frac>1e-3 {frac:.2e}'
        if ref_perturbed is not None:
            f_err, f_frac = self._stats(ref_perturbed, ref)
            self.floor.append((f_err, f_frac))
        raw, above = argmax_flips(got, ref, f_err)
        msg += f' argmax flips raw {raw}, at margin > {2 * max(f_err or 0.0, NORTH_STAR):.1e}: {above}'
        if ref_perturbed is not None:
            r_raw, r_above = argmax_flips(ref_perturbed, ref, f_err)
            self.ref_flips += r_raw
            msg += f' | ref vs ref(1e-6 input noise) max-abs {f_err:.2e} frac>1e-3 {f_frac:.2e} flips raw {r_raw}'
        print(msg)
        self.raw_flips += raw
        self.flips_above_floor += above
        self.ours.append((err, frac))
        self.frames.append(frame)
        top2 = ref.topk(2, dim=0)[0]
        flipped = got.argmax(0) != ref.argmax(0)
        self.pending.append((frame, (top2[0] - top2[1])[flipped]))  # reference margins of the flipped pixels

    def finish(self):
        fl_err = max([e for e, _ in self.floor] + [0.0])
        bound = NORTH_STAR if self.strict else max(NORTH_STAR, 10 * fl_err)
        ours_err = max(e for e, _ in self.ours)
        print(f'{self.tag}: clip max-abs {ours_err:.2e} (reference self-drift {fl_err:.2e}, bound {bound:.1e}); '
              f'top-k selections differing from the reference: {self.flips} (all explained near-ties), first discrete '
              f'difference at frame {self.first_flip_frame}; argmax flips raw {self.raw_flips} '
              f'(reference vs itself: {self.ref_flips}), at margin > 2 x max(frame floor, 1e-3): {self.flips_above_floor}')
        for t, (e, _) in zip(self.frames, self.ours):
            assert e <= bound, f'{self.tag} frame {t}: max-abs {e:.2e} above the bound {bound:.1e}'
            if e > NORTH_STAR:
                assert self.first_flip_frame is not None and t >= self.first_flip_frame, \
                    (f'{self.tag} frame {t}: error {e:.2e} above {NORTH_STAR:.0e} without a differing discrete decision '
                     'at or before this frame')
        for t, margins in self.pending:
            decisive = int((margins > 2 * bound).sum())
            assert decisive == 0, (f'{self.tag} frame {t}: {decisive} argmax flips at pixels whose reference margin '
                                   f'exceeds 2 x {bound:.1e}')
        return dict(max_abs=ours_err, floor=fl_err, bound=bound, raw_flips=self.raw_flips, ref_flips=self.ref_flips,
                    flips_above_floor=self.flips_above_floor, topk_flips=self.flips)


def _cmp(name, got, ref, tol, worst):
    got = got.detach().float().cpu()
    e = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item()) if ref.numel() else 0.0
    worst[name] = max(worst.get(name, 0.0), e)
    assert e <= tol, (name, e)


def _compare_banks(tag, hmem, omem, worst):
    """keys / selections bit-exact (they are copies: equality <=> same tokens in the same order),
    computed quantities (prototype values & shrinkage, usage counters) to 1e-5"""
    stores = [('work', hmem.work_mem, omem.work)]
    if omem.long_term and omem.long.buckets:
        stores.append(('long', hmem.long_mem, omem.long))
    for sname, hs, os_ in stores:
        assert sorted(hs.buckets) == sorted(os_.buckets), (tag, sname, hs.buckets, os_.buckets)
        hk, hshr = hs.key, hs.shrinkage
        for b in os_.buckets:
            assert hs.size(b) == os_.size(b), (tag, sname, b, hs.size(b), os_.size(b))
            assert torch.equal(hk[b].cpu(), os_.k[b]), f'{tag}: {sname} bucket {b}: key bank differs (different tokens kept / chosen)'
            _cmp(f'{sname}.shrinkage', hshr[b], os_.s[b], 1e-5, worst)
            if os_.keep_selection:
                assert torch.equal(hs.selection[b].cpu(), os_.e[b]), (tag, sname, b, 'selection')
            if os_.keep_usage:
                use, life = hs.usage_arenas(b)
                _cmp(f'{sname}.use_cnt', use[:hs.size(b)], os_.use[b], 1e-5, worst)
                _cmp(f'{sname}.life_cnt', life[:hs.size(b)], os_.life[b], 1e-6, worst)
        hv = hs.value
        for o, v in os_.v.items():
            _cmp(f'{sname}.value', hv[o], v, 1e-5, worst)


def teacher_forced_memory(P, sc: Dict, device, frames: int = None) -> Dict[str, float]:
    """see the module docstring (3).  Returns the worst relative error per compared quantity plus
    event counters."""
    from deva.inference.memory_manager import MemoryManager
    cfg = synth.base_config(**sc['cfg'])
    sc = dict(sc, frames=frames or sc['frames'])
    hmem = MemoryManager(cfg)
    worst: Dict[str, float] = {}
    events = dict(reads=0, adds=0, consolidations=0, evictions=0, tie_swapped_queries=0)
    state = dict(core=None, frame=-1)

    def install(core):
        omem = core.memory
        match0, add0 = omem.match, omem.add

        def match(key, selection):
            tag = f'frame {core.curr_ti}'
            with OracleTap() as otap:
                ro_o = match0(key, selection)
            with ReadTap() as htap:
                ro_h = hmem.match_memory(key.to(device), selection.to(device))
            assert len(otap.reads) == len(htap.reads)
            for bi, (hr, orr) in enumerate(zip(htap.reads, otap.reads)):
                assert torch.equal(hr['mk'], orr['mk']) and torch.equal(hr['qk'], orr['qk'])
                n_flip, excess, slack = explain_flips(f'{tag} bucket#{bi}', hr, orr)
                # identical inputs: a differing set is only acceptable at an fp32 tie
                assert excess <= slack, f'{tag}: top-k selection differs from the reference beyond a tie'
                events['tie_swapped_queries'] += n_flip
            for o, r in ro_o.items():
                _cmp('readout', ro_h[o], r, 1e-5 if events['tie_swapped_queries'] == 0 else 1e-2, worst)
            events['reads'] += 1
            _compare_banks(tag + ' after read', hmem, omem, worst)
            return ro_o

        def add(key, shrinkage, value, objects, selection):
            tag = f'frame {core.curr_ti} add'
            long_before = {b: omem.long.size(b) for b in omem.long.buckets} if omem.long_term else {}
            add0(key, shrinkage, value, objects, selection)
            hmem.add_memory(key.to(device), shrinkage.to(device), value.to(device), list(objects),
                            selection=selection.to(device))
            events['adds'] += 1
            if omem.long_term:
                for b in omem.long.buckets:
                    grown = omem.long.size(b) - long_before.get(b, 0)
                    if b not in long_before or grown != 0:
                        events['consolidations'] += 1
                        if grown < cfg['num_prototypes']:
                            events['evictions'] += 1
            _compare_banks(tag, hmem, omem, worst)

        omem.match, omem.add = match, add
        return core

    import scenarios
    scenarios.run_scenario(lambda c: install(O.OracleCore(P, c)), sc)
    worst.update({k: float(v) for k, v in events.items()})
    return worst
